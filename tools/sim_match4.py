"""The engine of k_match4 (szl_kernels_match2.hip, SZL_ENGINE_TEXT) in tools/wavesim.py, with its C++ fetch emulated in Python —
the baseline the new engine's instruction counts are compared with (same tile, same interleaving).  The fetch's cost is booked
from the compiled ISA (llvm-objdump of k_match4<false>): per main-loop iteration ~40 v_mov around the asm statement, and per
context ~10 VALU to retire + ~25 to allocate, ~35 SALU."""
import argparse, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import wavesim as W
from sim_match9 import base_of, pack

B_HIST, B_TAIL, MAX_DIST, TILE = 32512, 264, 32506, 21504
DATA_BYTES = B_HIST + TILE + B_TAIL + 8


def old_text():
    src = open(os.path.join(ROOT, "sharpziplib_amd", "csrc", "szl_kernels_match2.hip")).read().split("\n")
    a = next(i for i, l in enumerate(src) if l.startswith("#define SZL_Q_ISSUE"))
    b = next(i for i, l in enumerate(src) if l.startswith("struct WalkCtx"))
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "h.h"), "w").write("\n".join(src[a:b]))
        open(os.path.join(td, "t.c"), "w").write('#include <stdio.h>\n#include "h.h"\nint main(void){ fputs(SZL_ENGINE_TEXT(SZL_Q_FINISH, SZL_Q_FINISH_LAST, SZL_V_COMPLETE, "", "", ""), stdout); return 0; }\n')
        subprocess.check_call(["gcc", os.path.join(td, "t.c"), "-o", os.path.join(td, "t")])
        return subprocess.check_output([os.path.join(td, "t")]).decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="enwik"); ap.add_argument("--kib", type=int, default=96); ap.add_argument("--tile", type=int, default=1)
    ap.add_argument("--level", type=int, default=6); ap.add_argument("--fth", type=int, default=32); ap.add_argument("--vth", type=int, default=2)
    ap.add_argument("--qkeep", type=int, default=64); ap.add_argument("--vkeep", type=int, default=2); ap.add_argument("--slice", type=int, default=128)
    ap.add_argument("--waves", type=int, default=16)
    a = ap.parse_args()
    import oracle_ffi as O
    from sharpziplib_amd import corpus
    n = a.kib << 10
    data = corpus.generate(a.kind, {"enwik": 0xE9, "logs": 0x106}[a.kind], 0, n)
    model = O.Model(data, a.level); P = model.P
    t0 = a.tile * TILE; tlen = min(TILE, n - t0); dlo = t0 - B_HIST
    lds = np.zeros(163840, dtype=np.uint8)
    pos = np.arange(dlo, dlo + DATA_BYTES); ok = (pos >= 0) & (pos < n)
    lds[:DATA_BYTES][ok] = data[pos[ok]]
    nl = B_HIST + TILE
    pos = np.arange(dlo, dlo + nl); ok = (pos >= 0) & (pos < t0 + tlen)
    lk = np.zeros(nl, dtype=np.uint16); lk[ok] = model.link[pos[ok]]; lk[lk == 0] = 0xFFFF
    lds[DATA_BYTES:DATA_BYTES + 2 * nl] = lk.view(np.uint8)
    slink = lk
    base_lo, base_hi = base_of(t0), base_of(t0 + tlen - 1)
    sw = (1 << 30) if base_lo == base_hi else (base_lo + 65273) - t0
    sw = min(sw, TILE)
    basem_lo, basem_hi = base_lo - dlo, base_hi - dlo
    rem0 = min(n - t0, 1 << 24)
    prog = W.Program(old_text())
    print("engine: %d instructions" % len(prog.ins))
    mt2 = np.full(TILE + 64, 0xDEADBEEF, dtype=np.uint32); mtq = np.full(TILE + 64, 0xDEADBEEF, dtype=np.uint32)
    counter = [0]
    lanes = np.arange(64)
    vn = [x + c for c in "AB" for x in ["p", "cl", "best", "left", "off", "pb", "res2", "resq", "mincl", "cap", "nice", "t0", "t1", "t2", "t3", "t4", "t5", "t6", "t7"]]
    SNAP = P.max_chain - (P.max_chain >> 2)
    waves = []
    for wi in range(a.waves):
        v = {k: 0 for k in vn}
        for c in "AB":
            v["cl" + c] = B_HIST; v["best" + c] = 2; v["cap" + c] = 258; v["nice" + c] = P.nice
        s = dict(qA=0, vA=0, dA=0, qB=0, vB=0, dB=0, mA=0, mB=0, sc=0, cm=0, sv=0, n0=0, n1=0, n2=0, lbase=DATA_BYTES, dbase=0, pbase=B_HIST,
                 dbm1=W.M32, pbm1=B_HIST - 1, bhist=B_HIST, snap=SNAP, bexit=128 - a.fth, vth=a.vth, qkeep=a.qkeep, vkeep=a.vkeep)
        w = W.Wave(prog, lds, v, s, {})
        w.wnext = w.wend = 0; w.exhausted = False; w.fetch_valu = 0; w.fetch_salu = 0; w.fetch_lanes = 0; w.fetch_lds = 0; w.loops = 0
        w.tail_steps0 = 0; w.pc_hist = {}
        waves.append(w)

    def fetch(w, C):
        q, vv, dm = w.s["q" + C], w.s["v" + C], w.s["d" + C]
        if dm:
            em = W._mask_to_bool(dm)
            p = w.v["p" + C][em]; r2 = w.v["res2" + C][em]; rq = w.v["resq" + C][em]
            e = pack(r2, rq); mt2[p] = e
            c2 = (e >> 24) == 2
            mtq[p[c2]] = rq[c2]
            w.fetch_valu += 10
        w.s["d" + C] = 0
        w.fetch_salu += 12
        if w.exhausted:
            return
        idle = ~(q | vv) & W.M64
        ni = bin(idle).count("1")
        if ni == 0:
            return
        if w.wnext >= w.wend:
            base = counter[0]; counter[0] += a.slice
            w.wnext = min(base, tlen); w.wend = min(base + a.slice, tlen)
            w.fetch_salu += 10; w.fetch_lds += 1
            if w.wnext >= w.wend:
                w.exhausted = True
                return
        im = W._mask_to_bool(idle)
        rank = np.cumsum(im) - im
        w.fetch_valu += 25; w.fetch_salu += 23; w.fetch_lds += 2
        tov = np.zeros(64, bool)
        for l in lanes[im]:
            p = w.wnext + int(rank[l])
            if p >= w.wend:
                continue
            w.fetch_lanes += 1
            w.v["p" + C][l] = p
            rem = rem0 - p
            w.v["res2" + C][l] = 0; w.v["resq" + C][l] = 0
            ok = rem >= 3
            if ok:
                pl = p + B_HIST
                l0 = int(slink[pl])
                basem = basem_hi if p >= sw else basem_lo
                firstmin = max(pl - MAX_DIST, basem)
                c = pl - l0
                ok = c >= firstmin
                if ok:
                    w.v["cl" + C][l] = c
                    w.v["mincl" + C][l] = max(pl - (MAX_DIST - 1), basem) & W.M32
                    w.v["cap" + C][l] = min(rem, 258); w.v["nice" + C][l] = min(rem, P.nice)
                    w.v["best" + C][l] = 2; w.v["left" + C][l] = P.max_chain - 1
                    w.v["pb" + C][l] = (int(lds[pl + 2]) << 8) | int(lds[pl + 1])
                    w.v["off" + C][l] = 0
                    tov[l] = True
            if not ok:
                mt2[p] = 0
        w.s["v" + C] = vv | W._bool_to_mask(tov)
        w.wnext = min(w.wnext + ni, w.wend)

    # drive: each wave alternates fetch (python) and one engine call (interpreted until it runs off the end)
    main_cnt = {}
    live = list(waves)
    while live:
        for w in list(live):
            if w.done or w.pc == 0:
                fetch(w, "A"); fetch(w, "B")
                w.fetch_valu += 40; w.loops += 1
                if (w.s["qA"] | w.s["vA"] | w.s["qB"] | w.s["vB"]) == 0:
                    if w.exhausted:
                        live.remove(w); continue
                    else:
                        continue
                w.s["bexit"] = 0 if w.exhausted else 128 - a.fth
                w.pc = 0; w.done = False
            for _ in range(300):
                if w.done:
                    break
                if w.exhausted:
                    w.tail_steps0 += 1
                w.pc_hist[w.pc] = w.pc_hist.get(w.pc, 0) + 1
                if not w.exhausted:
                    op0 = prog.ins[w.pc][0]
                    kk = "valu" if op0.startswith("v_") else ("lds" if op0.startswith("ds_") else "salu")
                    main_cnt[kk] = main_cnt.get(kk, 0) + 1
                w.step()
    want = pack(model.m2[t0:t0 + tlen], model.mq[t0:t0 + tlen])
    bad = np.where(mt2[:tlen] != want)[0]
    print("positions %d mismatches %d" % (tlen, bad.size))
    cnt, ln = W.merge_counts(waves)
    va = sum(v for k, v in cnt.items() if k[1] == "valu"); sa = sum(v for k, v in cnt.items() if k[1] in ("salu", "branch")); ld = sum(v for k, v in cnt.items() if k[1] == "lds")
    fv = sum(w.fetch_valu for w in waves); fs = sum(w.fetch_salu for w in waves); fl = sum(w.fetch_lds for w in waves)
    print("engine: valu %.3f salu %.3f lds %.3f per position; fetch (booked): valu %.3f salu %.3f lds %.3f; lanes per fetch visit %.1f" % (
        va / tlen, sa / tlen, ld / tlen, fv / tlen, fs / tlen, fl / tlen, sum(w.fetch_lanes for w in waves) / max(1, sum(w.loops for w in waves)) / 2))
    print("total: valu %.2f salu %.2f lds %.2f  = %.2f" % ((va + fv) / tlen, (sa + fs) / tlen, (ld + fl) / tlen, (va + fv + sa + fs + ld + fl) / tlen))
    # per-region counts (regions = label-delimited blocks of the text)
    marks = sorted((pc, lab) for lab, pcs in prog.labels.items() for pc in pcs)
    import bisect
    reg = {}
    for w in waves:
        for pc, c in w.pc_hist.items():
            i = bisect.bisect_right([m[0] for m in marks], pc) - 1
            lab = marks[i][1] if i >= 0 else "start"
            op = prog.ins[pc][0]
            cls = "valu" if op.startswith("v_") else ("lds" if op.startswith("ds_") else "salu")
            reg.setdefault(lab, {}).setdefault(cls, 0)
            reg[lab][cls] += c
    for lab in sorted(reg, key=lambda x: int(x) if x.isdigit() else -1):
        r = reg[lab]
        print("  block %-5s valu %.3f salu %.3f lds %.3f" % (lab, r.get("valu", 0) / tlen, r.get("salu", 0) / tlen, r.get("lds", 0) / tlen))
    print("main phase engine only: valu %.2f salu %.2f lds %.2f (+ fetch as booked)" % tuple(main_cnt.get(k, 0) / tlen for k in ("valu", "salu", "lds")))
    ts = [w.tail_steps0 for w in waves]
    print("tail instructions per wave: min %d mean %d max %d; share of engine instructions %.2f" % (min(ts), sum(ts) // len(ts), max(ts), sum(ts) / (va + sa + ld)))


if __name__ == "__main__":
    main()
