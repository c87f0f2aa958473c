"""Run k_match9's instruction text (sharpziplib_amd/csrc/szl_match9_asm.h) on the CPU through tools/wavesim.py: one tile staged
the way the kernel stages it, 16 interleaved wavefronts, tables diffed against oracle/szl_model.c, instruction counts per
position and lane occupancy per phase printed.

  python tools/sim_match9.py [--kind enwik] [--mib 1] [--tile 1] [--tlen 21504] [--level 6] [--nq 3] [--waves 16]
                             [--fth 24] [--vth 2] [--wth 2] [--wkeep 2] [--qkeep 64] [--qkeept 1] [--slice 128] [--abs0 0]
"""
import argparse, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import wavesim as W

B_HIST, B_TAIL, MAX_DIST = 32512, 264, 32506
D, M64 = 16, (1 << 64) - 1


def engine_text(nq, header="szl_match9_asm.h", macro="SZL9_TEXT", defs=()):
    src = '#include <stdio.h>\n#include "%s"\nint main(void){ fputs(%s, stdout); return 0; }\n' % (header, macro)
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c"); exe = os.path.join(td, "t")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-DSZL9_NQ=%d" % nq, *["-D" + d for d in defs], "-I", os.path.join(ROOT, "sharpziplib_amd", "csrc"), c, "-o", exe])
        return subprocess.check_output([exe]).decode()


def base_of(s_abs):
    idx = s_abs + 1
    if idx <= 65273:
        return 0
    return ((idx - 65273 + 32767) >> 15) << 15


def stage_tile(data, link, t0, tlen, seg_end, tile_cap):
    """LDS image of a tile (the kernel's b9_stage_window): counter, bytes, links"""
    data_bytes = B_HIST + tile_cap + B_TAIL + 8
    lb = D + data_bytes
    lds = np.zeros(163840, dtype=np.uint8)
    dlo = t0 - B_HIST
    pos = np.arange(dlo, dlo + data_bytes)
    ok = (pos >= 0) & (pos < seg_end)
    lds[D:D + data_bytes][ok] = data[pos[ok]]
    nl = B_HIST + tile_cap
    pos = np.arange(dlo, dlo + nl)
    ok = (pos >= 0) & (pos < t0 + tlen)
    lk = np.zeros(nl, dtype=np.uint16)
    lk[ok] = link[pos[ok]]
    lk[lk == 0] = 0xFFFF
    lds[lb:lb + 2 * nl] = lk.view(np.uint8)
    return lds, lb


def tile_consts(abs0, t0, tlen, seg_end, tile_cap):
    dlo = t0 - B_HIST
    base_lo, base_hi = base_of(abs0 + t0), base_of(abs0 + t0 + tlen - 1)
    sw = (1 << 30) if base_lo == base_hi else (base_lo + 65273) - abs0 - t0
    sw = min(sw, tile_cap)
    return dict(rem0=min(seg_end - t0, 1 << 24), sw=sw, bmlo=(base_lo - abs0 - dlo) & W.M32, bmhi=(base_hi - abs0 - dlo) & W.M32)


def pack(r2, rq):
    code = np.where(rq == r2, 0, np.where(rq == 0, 1, 2)).astype(np.uint32)
    return (r2 & 0x1FF) | ((r2 >> 16) << 9) | (code << 24)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="enwik"); ap.add_argument("--kib", type=int, default=128); ap.add_argument("--tile", type=int, default=1)
    ap.add_argument("--tlen", type=int, default=21504); ap.add_argument("--level", type=int, default=6); ap.add_argument("--nq", type=int, default=3)
    ap.add_argument("--waves", type=int, default=16); ap.add_argument("--fth", type=int, default=16); ap.add_argument("--vth", type=int, default=2)
    ap.add_argument("--wth", type=int, default=2); ap.add_argument("--wkeep", type=int, default=2); ap.add_argument("--qkeep", type=int, default=64)
    ap.add_argument("--qkeept", type=int, default=128); ap.add_argument("--vtht", type=int, default=1); ap.add_argument("--ktail", type=int, default=2); ap.add_argument("--slice", type=int, default=128); ap.add_argument("--abs0", type=int, default=0)
    ap.add_argument("--notail", action="store_true"); ap.add_argument("--strategy", type=int, default=0); ap.add_argument("--quantum", type=int, default=300)
    ap.add_argument("--tailp", type=int, default=2); ap.add_argument("--mth", type=int, default=64); ap.add_argument("--ktail1", type=int, default=4)
    ap.add_argument("--guide", type=int, default=8192, help="within this many positions of the tile's end a fetch takes its free lanes' worth of positions")
    ap.add_argument("--adapt", action="store_true", help="form 1 of the text (SZL9_V 1: the first filter byte follows the last failed compare)")
    ap.add_argument("--cut", type=int, default=0, help="segment ends this many bytes before the end of the generated data's last tile (lookahead clamps)")
    a = ap.parse_args()
    import oracle_ffi as O
    from sharpziplib_amd import corpus
    n = a.kib << 10
    seed = {"enwik": 0xE9, "logs": 0x106, "dickens": 0xD1CE}.get(a.kind, 1)
    if a.kind == "zeros":
        data = np.zeros(n, np.uint8)
    else:
        data = corpus.generate(a.kind, seed, 0, n)
    if a.cut:
        data = data[:n - a.cut]; n = data.size
    t = time.time()
    model = O.Model(data, a.level, strategy=a.strategy)
    P = model.P
    tile_cap = 21504
    t0 = a.tile * a.tlen
    tlen = min(a.tlen, n - t0)
    assert tlen > 0, "tile beyond the data"
    lds, lb = stage_tile(data, model.link, t0, tlen, n, tile_cap)
    assert lb == 54304
    K = tile_consts(a.abs0, t0, tlen, n, tile_cap)
    text = engine_text(a.nq, defs=["SZL9_V=1"] if a.adapt else [])
    prog = W.Program(text)
    print("program: %d instructions; model %.1fs" % (len(prog.ins), time.time() - t), flush=True)
    mt2 = np.full(B_HIST + tile_cap + 64, 0xDEADBEEF, dtype=np.uint32)
    mtq = np.full(B_HIST + tile_cap + 64, 0xDEADBEEF, dtype=np.uint32)
    vnames = ["vzero", "vslice"] + [x + c for c in "AB" for x in
                                   ["pl", "cb", "kk", "mincb", "left", "pb", "best", "off", "cap", "nice", "res2", "resq", "p0", "p1", "p2", "p3", "hop", "kd",
                                    "t0", "t1", "t2", "t3", "t4", "t5", "t6", "t7"]]
    waves = []
    for w in range(a.waves):
        v = {k: 0 for k in vnames}
        v["vslice"] = a.slice
        for c in "AB":
            v["best" + c] = 2; v["cap" + c] = 258; v["nice" + c] = P.nice
        s = dict(qA=0, vA=0, wA=0, dA=0, mA=0, cA=0, qB=0, vB=0, wB=0, dB=0, mB=0, cB=0, sc=0, cm=0, sa=0, sv=0, n0=0, n1=0, n2=0, f0=0, f1=0, f2=0,
                 wnext=0, wend=0, exh=0, tlen=tlen, slice=a.slice, rem0=K["rem0"], sw=K["sw"] & W.M32, bmlo=K["bmlo"], bmhi=K["bmhi"],
                 nicel=P.nice, chainm2=(P.max_chain - 2) & W.M32, snapm1=(P.max_chain - (P.max_chain >> 2) - 1) & W.M32, bexit=64 - a.fth, vth=a.vth, wth=a.wth,
                 wkeep=a.wkeep, qkeep=a.qkeep, qkeept=a.qkeept, vtht=a.vtht, ktail=a.ktail, kt=0, texh=0, stratm=0 if a.strategy == 2 else M64, mt2b=0, mtqb=0,
                 tailp=a.tailp, mth=a.mth & W.M32, ktail1=a.ktail1, vtht1=1, wscr=162368 + 64 * w, guide=(tlen - a.guide) & W.M32)
        waves.append(W.Wave(prog, lds, v, s, {"mt2b": mt2, "mtqb": mtq}))
        waves[-1].tail_flag = None if a.notail else "exh"
    t = time.time()
    W.run_workgroup(waves, quantum=a.quantum)
    print("simulated in %.1fs" % (time.time() - t))
    # ---- compare
    want2 = model.m2[t0:t0 + tlen]; wantq = model.mq[t0:t0 + tlen]
    wante = pack(want2, wantq)
    got = mt2[B_HIST:B_HIST + tlen]
    bad = np.where(got != wante)[0]
    code2 = np.where((wante >> 24) == 2)[0]
    badq = code2[mtq[B_HIST:B_HIST + tlen][code2] != wantq[code2]]
    stray = np.where(mt2[:B_HIST] != 0xDEADBEEF)[0].size + np.where(mt2[B_HIST + tlen:] != 0xDEADBEEF)[0].size
    print("positions %d: m2 mismatches %d, mq mismatches %d (of %d), stray stores %d" % (tlen, bad.size, badq.size, code2.size, stray))
    for i in bad[:8]:
        print("  p=%d got %08x want %08x (m2 %08x mq %08x)" % (i, got[i], wante[i], want2[i], wantq[i]))
    for i in badq[:4]:
        print("  p=%d mq got %08x want %08x" % (i, mtq[B_HIST + i], wantq[i]))
    cnt, lanes = W.merge_counts(waves)
    phases = sorted({k[0] for k in cnt})
    tot = {}
    print("%-10s %10s %10s %10s %8s   (wave-instructions per position; lane occupancy of VALU)" % ("phase", "valu", "salu+br", "lds", "occ"))
    for ph in phases:
        va = cnt.get((ph, "valu"), 0); sa = cnt.get((ph, "salu"), 0) + cnt.get((ph, "branch"), 0); ld = cnt.get((ph, "lds"), 0)
        occ = lanes.get((ph, "valu"), 0) / (64.0 * va) if va else 0
        print("%-10s %10.3f %10.3f %10.3f %8.2f" % (ph or "-", va / tlen, sa / tlen, ld / tlen, occ))
        for k, v in (("valu", va), ("salu", sa), ("lds", ld)):
            tot[k] = tot.get(k, 0) + v
    print("%-10s %10.3f %10.3f %10.3f   vmem %.3f  unaligned-lds %d" % ("total", tot["valu"] / tlen, tot["salu"] / tlen, tot["lds"] / tlen,
                                                                       sum(v for k, v in cnt.items() if k[1] == "vmem") / tlen, cnt.get(("", "lds_unaligned"), 0)))
    ts = [w.tail_steps for w in waves]
    print("tail (after the tile is handed out): instructions per wave min %d / mean %d / max %d" % (min(ts), sum(ts) // len(ts), max(ts)))
    main = sum(v for k, v in cnt.items() if not k[0].startswith("tail") and k[1] in ("valu", "salu", "branch", "lds", "vmem"))
    print("main phase: %.2f wave-instructions per position (valu %.2f)" % (main / tlen, sum(v for k, v in cnt.items() if not k[0].startswith("tail") and k[1] == "valu") / tlen))
    return 0 if (bad.size == 0 and badq.size == 0 and stray == 0) else 1


if __name__ == "__main__":
    sys.exit(main())
