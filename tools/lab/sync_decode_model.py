"""Design experiment (CPU only): would sub-spans of a block, decoded by the 64 lanes from guessed bit offsets and stitched where a chain meets its successor, beat the speculative round of k_inflate?  Counts SIMD decode steps per input bit for sub-span lengths s and re-decode rounds R on zlib level-6 text (python tools/lab/sync_decode_model.py [kind]); DESIGN.md §8.2 has the table and the verdict."""
import sys, zlib, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sharpziplib_amd import corpus as C
kind = sys.argv[1] if len(sys.argv) > 1 else 'enwik'
d = C.generate(kind, 0xE9, 0, 1 << 20).tobytes()
co = zlib.compressobj(6, zlib.DEFLATED, -15); comp = co.compress(d) + co.flush()
bits = np.unpackbits(np.frombuffer(comp, np.uint8), bitorder='little')
nb = len(bits)
def rd(p, n):
    v = 0
    for i in range(n): v |= int(bits[p + i]) << i
    return v
LB = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LE = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DE = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
def mk(lens):
    cnt = [0]*16
    for l in lens: cnt[l] += 1
    cnt[0] = 0; code = 0; nxt = [0]*16
    for l in range(1, 16): code = (code + cnt[l-1]) << 1; nxt[l] = code
    t = {}
    for s, l in enumerate(lens):
        if l: t[(l, nxt[l])] = s; nxt[l] += 1
    return t
def dec(t, p):
    c = 0
    for l in range(1, 16):
        if p + l > nb: return None, 0
        c = (c << 1) | int(bits[p + l - 1])
        if (l, c) in t: return t[(l, c)], l
    return None, 0
p = 0; blocks = []
while True:
    fin = rd(p, 1); ty = rd(p + 1, 2); p += 3
    assert ty == 2, ty
    hl = rd(p, 5) + 257; hd = rd(p + 5, 5) + 1; hc = rd(p + 10, 4) + 4; p += 14
    order = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]; ml = [0]*19
    for i in range(hc): ml[order[i]] = rd(p, 3); p += 3
    mt = mk(ml); lens = []
    while len(lens) < hl + hd:
        s, l = dec(mt, p); p += l
        if s < 16: lens.append(s)
        elif s == 16: lens += [lens[-1]] * (3 + rd(p, 2)); p += 2
        elif s == 17: lens += [0] * (3 + rd(p, 3)); p += 3
        else: lens += [0] * (11 + rd(p, 7)); p += 7
    lt = mk(lens[:hl]); dt = mk(lens[hl:])
    start = p
    # token length at every bit offset of the block (until the true EOB is found by the true chain)
    def tok(q):
        s, l = dec(lt, q)
        if s is None: return 0, 0
        if s < 256: return l, 1
        if s == 256: return -l, 0
        if s > 285: return 0, 0
        n = l + LE[s - 257]; ln = LB[s - 257] + rd(q + l, LE[s - 257]) if q + n <= nb else 0
        ds, dl = dec(dt, q + n)
        if ds is None or ds > 29: return 0, 0
        return n + dl + DE[ds], ln
    q = start; true = []
    while True:
        n, ob = tok(q)
        if n < 0: q -= n; break
        true.append(q); q += n
    blocks.append((start, q, lt, dt, tok, true))
    p = q
    if fin: break
    if len(blocks) >= 4: break
print(kind, 'blocks', len(blocks), [(b[1] - b[0]) // 8 for b in blocks], 'bytes; tokens', [len(b[5]) for b in blocks])
import random
random.seed(1)
for s in (64, 96, 128, 192, 256):
    for R in (1, 2, 3, 4):
        tot_steps = 0; tot_bits = 0; spans = 0
        for (start, end, lt, dt, tok, true) in blocks[:2]:
            memo = {}
            def T(q):
                if q not in memo: memo[q] = tok(q)
                return memo[q]
            P = start
            while P + 64 * s + 128 < end:
                # lanes
                entry = [P + i * s for i in range(64)]
                exitp = [None] * 64; used = [None] * 64
                steps = 0
                for it in range(R + 1):
                    mx = 0
                    newexit = list(exitp)
                    for i in range(64):
                        e = entry[i]
                        if used[i] == e: continue
                        q = e; n = 0; lim = P + (i + 1) * s
                        while q < lim and n < 64:
                            k, _ = T(q)
                            if k <= 0: break
                            q += k; n += 1
                        newexit[i] = q if (q >= lim) else -1   # -1: stopped (invalid / EOB / cap)
                        used[i] = e; mx = max(mx, n)
                    exitp = newexit; steps += mx
                    for i in range(1, 64):
                        entry[i] = exitp[i - 1] if exitp[i - 1] is not None and exitp[i - 1] > 0 else entry[i]
                # valid prefix
                v = 1
                while v < 64 and exitp[v - 1] > 0 and used[v] == exitp[v - 1]: v += 1
                newP = exitp[v - 1] if exitp[v - 1] > 0 else None
                if newP is None: break
                tot_steps += steps; tot_bits += newP - P; spans += 1
                P = newP
        print(f's={s:3d} R={R}: spans {spans:4d}  avg valid bits/span {tot_bits / max(spans, 1):7.0f} of {64 * s}   serial steps/span {tot_steps / max(spans, 1):6.1f}   bits per step {tot_bits / max(tot_steps, 1):6.1f}')
