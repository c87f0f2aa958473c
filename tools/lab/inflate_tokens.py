"""Tiny raw-DEFLATE reader that returns the token list (bring-up helper): [(pos, 'L', byte) | (pos, 'M', len, dist) | (pos, 'B', type, last) | (pos, 'S', n)]."""
LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data):
        self.d, self.p = data, 0
    def get(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[self.p >> 3] >> (self.p & 7)) & 1) << i
            self.p += 1
        return v


def table(lens):
    codes, code, out = {}, 0, {}
    bl = [0] * 16
    for l in lens:
        bl[l] += 1
    bl[0] = 0
    nxt = [0] * 16
    for b in range(1, 16):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    for s, l in enumerate(lens):
        if l:
            out[(l, nxt[l])] = s
            nxt[l] += 1
    return out


def sym(br, t):
    c, l = 0, 0
    while True:
        c = (c << 1) | br.get(1); l += 1
        if (l, c) in t:
            return t[(l, c)]
        if l > 15:
            raise ValueError("bad code")


def tokens(data, limit=None):
    br, out, pos = Bits(data), [], 0
    while True:
        last, typ = br.get(1), br.get(2)
        out.append((pos, 'B', typ, last))
        if typ == 0:
            br.p = (br.p + 7) & ~7
            n = br.get(16); br.get(16)
            br.p += 8 * n
            out.append((pos, 'S', n)); pos += n
        else:
            if typ == 1:
                ll = table([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8); dd = table([5] * 30)
            else:
                hl, hd, hc = br.get(5) + 257, br.get(5) + 1, br.get(4) + 4
                cl = [0] * 19
                for i in range(hc):
                    cl[ORDER[i]] = br.get(3)
                ct, lens = table(cl), []
                while len(lens) < hl + hd:
                    s = sym(br, ct)
                    if s < 16: lens.append(s)
                    elif s == 16: lens += [lens[-1]] * (3 + br.get(2))
                    elif s == 17: lens += [0] * (3 + br.get(3))
                    else: lens += [0] * (11 + br.get(7))
                ll, dd = table(lens[:hl]), table(lens[hl:])
            while True:
                s = sym(br, ll)
                if s < 256:
                    out.append((pos, 'L', s)); pos += 1
                elif s == 256:
                    break
                else:
                    ln = LBASE[s - 257] + br.get(LEXT[s - 257])
                    ds = sym(br, dd)
                    dist = DBASE[ds] + br.get(DEXT[ds])
                    out.append((pos, 'M', ln, dist)); pos += ln
        if last or (limit and pos >= limit):
            return out
