"""gfxsim.phases — where stage B's instructions go, without a GPU: k_match9 on the interpreter with FULL-LENGTH tiles
(SZL_TILE_FLOOR=21504 keeps them on a small input — the steady state of a long stream), every executed instruction booked to the
`; @phase` marker of the assembly it stands behind (csrc/szl_match9_asm.h) and to its class.

    python tools/gfxsim/phases.py [bytes=215040] [level=6] [kind=enwik]

Calibration (round 4): a 2.9 MB text stream at level 6 — what the library does with default knobs — gives 12.78 VALU / 10.32 SALU /
4.05 LDS / 2.16 branch wave-instructions per position on the interpreter; rocprofv3's SQ_INSTS_* of the 1 GiB bench step
(profiles/r04/pmc_sq_1gib.json) say 12.82 / 10.35 / 4.07 / 2.16.  The counts are the hardware's; what the interpreter cannot say is
how long they take.  Test infrastructure only.
"""
import bisect
import collections
import os
import re
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(argv):
    from gfxsim import harness
    rt = harness.use(fast_probe=True)
    import oracle_ffi as O
    from sharpziplib_amd import corpus as C, _lib
    from sharpziplib_amd.batch import Engine
    n = int(argv[0]) if argv else 215040
    lv = int(argv[1]) if len(argv) > 1 else 6
    kind = argv[2] if len(argv) > 2 else "enwik"
    _lib.lib().szl_debug_set(b"SZL_TILE_FLOOR", 21504)
    e = Engine()
    data = C.generate(kind, 0xE9, 0, n)
    src = open(os.path.join(HERE, "_build", "szl_kernels_match9.s")).read().splitlines()
    marks = [(1, "prologue + staging")] + [(i + 1, re.sub(r".*@phase\s+", "", l).strip()) for i, l in enumerate(src) if "@phase" in l]
    mlines = [m[0] for m in marks]
    seen = collections.Counter()
    names = []
    loops = {1: "main", 2: "tail, two contexts", 3: "tail, one context"}
    for ln, nm in marks:
        seen[nm] += 1
        in_loop = nm in ("census", "quick", "classify", "verify1", "verify2", "complete")
        names.append("%s (%s)" % (nm, loops.get((seen[nm] - 1) % 3 + 1)) if in_loop else nm)
    cnt = collections.Counter()

    def tr(w, I):
        cnt[(names[bisect.bisect_right(mlines, I.line) - 1], I.cls)] += 1
    orig = rt.launch

    def launch(name, *a, **k):
        rt.trace = tr if "k_match9ILb0" in name else None
        return orig(name, *a, **k)
    rt.launch = launch
    t = time.time()
    r = e.deflate([data], level=lv)[0]
    print("%d bytes of %s, level %d: bytes == oracle: %s (%.0f s)" % (n, kind, lv, r.data == O.deflate(data, lv), time.time() - t))
    order = []
    for nm in names:
        if nm not in order:
            order.append(nm)
    tot = collections.Counter()
    print("wave-instructions per position\n%-32s %7s %7s %7s %7s %7s %7s" % ("phase", "valu", "salu", "lds", "branch", "other", "all"))
    for ph in order:
        row = {c: cnt[(ph, c)] / n for c in ("valu", "salu", "lds", "branch", "other", "vmem", "smem")}
        if sum(row.values()) == 0:
            continue
        for c, v in row.items():
            tot[c] += v
        print("%-32s %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f" % (ph, row["valu"], row["salu"], row["lds"], row["branch"], row["other"] + row["vmem"] + row["smem"], sum(row.values())))
    print("%-32s %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f" % ("TOTAL", tot["valu"], tot["salu"], tot["lds"], tot["branch"], tot["other"] + tot["vmem"] + tot["smem"], sum(tot.values())))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
