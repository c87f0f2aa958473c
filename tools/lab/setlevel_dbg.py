"""Scripted SetLevel scenarios against the oracle (bring-up helper for the function-switch paths): prints the first differing byte."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_ffi as O
from sharpziplib_amd import corpus as C
from sharpziplib_amd.deflater import Deflater


def run(name, level, ops, seed=1, nowrap=True):
    data = C.generate("enwik", seed, 0, 400000)
    d, o = Deflater(level, nowrap), O.Deflater(level, nowrap)
    got, ref = bytearray(), bytearray()
    buf = np.zeros(8192, np.uint8)
    pos = 0

    def drain():
        while True:
            k = d.Deflate(buf)
            if k <= 0:
                break
            got.extend(buf[:k].tobytes())
        while True:
            b = o.deflate(8192)
            if not b:
                break
            ref.extend(b)

    try:
        for op in ops:
            if op[0] == "in":
                c = data[pos:pos + op[1]]; pos += op[1]
                d.SetInput(c); o.set_input(c)
            elif op[0] == "lvl":
                d.SetLevel(op[1]); o.set_level(op[1])
            elif op[0] == "drain":
                drain()
            elif op[0] == "flush":
                d.Flush(); o.flush(); drain()
        d.Finish(); o.finish()
        while not d.IsFinished:
            k = d.Deflate(buf)
            if k <= 0:
                break
            got.extend(buf[:k].tobytes())
        while not o.finished:
            ref.extend(o.deflate(8192))
    except Exception as e:
        print("%-40s EXC %r" % (name, e)); return
    g, r = bytes(got), bytes(ref)
    if g == r:
        print("%-40s ok (%d bytes)" % (name, len(g)))
    else:
        k = next((i for i in range(min(len(g), len(r))) if g[i] != r[i]), min(len(g), len(r)))
        print("%-40s DIFF at %d of %d/%d  got %s ref %s" % (name, k, len(g), len(r), g[max(0,k-2):k + 6].hex(), r[max(0,k-2):k + 6].hex()))
        if nowrap and TOK:
            sys.path.insert(0, "tools")
            from inflate_tokens import tokens
            tg, tr = tokens(g + b"\0" * 8), tokens(r + b"\0" * 8)
            for i in range(min(len(tg), len(tr))):
                if tg[i] != tr[i]:
                    print("    first differing token #%d: got %s ref %s; before: %s" % (i, tg[i:i + 3], tr[i:i + 3], tr[max(0, i - 3):i]))
                    q = tr[i][0]
                    print("    data around: %r" % bytes(data[max(0, q - 8):q + 24]))
                    break


TOK = True
I, L, D, F = (lambda n: ("in", n)), (lambda l: ("lvl", l)), ("drain",), ("flush",)
run("slow>stored after flush", 6, [I(5000), D, F, L(0), I(3000), D])
run("stored>slow after flush", 0, [I(5000), D, F, L(6), I(3000), D])
run("stored>slow drained", 0, [I(5000), D, L(6), I(3000), D])
run("stored>slow drained, drain", 0, [I(5000), D, L(6), D, I(3000), D])
run("stored>fast drained", 0, [I(5000), D, L(2), I(3000), D])
run("slow>stored drained", 6, [I(5000), D, L(0), I(3000), D])
run("slow>stored drained, drain", 6, [I(5000), D, L(0), D, I(3000), D])
run("slow>stored short", 6, [I(100), D, L(0), I(3000), D])
run("fast>stored drained", 2, [I(5000), D, L(0), I(3000), D])
run("slow>stored>slow", 6, [I(5000), D, L(0), I(3000), D, L(6), I(4000), D])
run("slow>stored>slow no drain", 6, [I(5000), D, L(0), L(6), I(4000), D])
run("slow>stored,drain>slow", 6, [I(5000), D, L(0), D, L(6), I(4000), D])
run("stored big>slow", 0, [I(70000), D, I(50000), D, L(6), I(40000), D])
run("stored>slow>stored>fast", 0, [I(5000), D, L(6), I(30000), D, L(0), I(40000), D, L(3), I(50000), D])
run("zlib stored>slow", 0, [I(5000), D, L(6), I(3000), D], nowrap=False)
run("zlib slow>stored>fast", 6, [I(5000), D, L(0), I(3000), D, L(1), I(9000), D], nowrap=False)
run("stored full window>slow", 0, [I(65535), D, L(6), I(3000), D])
run("stored 65273>slow", 0, [I(65273), D, L(6), I(3000), D])
run("stored 65272>fast", 0, [I(65272), D, L(3), I(3000), D])
