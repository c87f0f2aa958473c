"""gfxsim.srcprof — a source-line profile of one kernel on the interpreter: which lines of the .hip file its executed instructions come from.

    python tools/gfxsim/srcprof.py inflate [bytes=500000]        the symbol pass of the chunk-parallel Inflater on one text member
    python tools/gfxsim/srcprof.py parse [bytes=262144]          k_spec_win on a text stream at level 6
    python tools/gfxsim/srcprof.py blocks [bytes=262144]         k_block_build

The unit is compiled a second time with -gline-tables-only (same code, plus .loc directives); instruction k of the interpreted assembly is
instruction k of that listing.  Counts are wave-instructions (the device's SQ_INSTS_*, tools/gfxsim/phases.py has the calibration); for a
kernel that is one wavefront's dependency chain (k_inflate) they are a fair picture of where its time goes.  Test infrastructure only.
"""
import bisect
import collections
import os
import re
import subprocess
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORK = {"inflate": ("szl_kernels_inflate", "k_inflateILb1ELi2ELb0", 500000), "parse": ("szl_kernels_parse", "k_spec_winILi32", 262144),
        "blocks": ("szl_kernels_block", "k_block_build", 262144)}


def listing_with_lines(unit):
    from gfxsim import asm
    out = os.path.join(HERE, "_build", unit + ".lines.s")
    src = os.path.join(ROOT, "sharpziplib_amd", "csrc", unit + ".hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-DSZL_LAB=0",
                               "-gline-tables-only", src, "-o", out], stderr=subprocess.DEVNULL)
    keep, skipping = [], False
    for l in open(out).read().splitlines():
        st = l.strip()
        if st.startswith(".section"):
            skipping = ".debug" in st
        elif st.startswith((".text", ".rodata", ".amdgpu_metadata", ".data")):
            skipping = False
        keep.append("" if skipping else l)                    # (blank lines keep the numbering)
    mod = asm.Module("\n".join(keep), unit + ".lines")
    locs = [(i + 1, int(m.group(2)), int(m.group(1))) for i, l in enumerate(keep) for m in [re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)] if m]
    files = {int(m.group(1)): (m.group(3) or m.group(2)) for l in keep for m in [re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)] if m}
    return mod, locs, files


def main(argv):
    what = argv[0] if argv else "inflate"
    unit, ksel, nbytes = WORK[what]
    if len(argv) > 1:
        nbytes = int(argv[1])
    from gfxsim import harness, suite
    rt = harness.use(fast_probe=True)
    mg, locs, files = listing_with_lines(unit)
    loc_lines = [a for a, _, _ in locs]
    mod = [m for m in rt.modules if m.name == unit + ".s"][0]
    assert len(mod.ins) == len(mg.ins), "the listing with line tables is not the interpreted code"
    idx_of_line = {I.line: k for k, I in enumerate(mod.ins)}
    cnt = collections.Counter()

    def tr(w, I):
        cnt[idx_of_line[I.line]] += 1
    orig = rt.launch

    def launch(name, *a, **k):
        rt.trace = tr if ksel in name else None
        return orig(name, *a, **k)
    rt.launch = launch
    import oracle_ffi as O
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd import corpus as C
    e = Engine()
    data = C.generate("enwik", 0xE9, 0, nbytes)
    if what == "inflate":
        m = O.deflate(data, 6)
        suite._knobs(SZL_INF_CHUNK_KIB=16, SZL_INF_PAR_MIN_KIB=64)
        (r, used), = e.inflate([m], [data.size])
        assert r.data == data.tobytes() and e._L.szl_engine_debug_par_jobs(e._h) >= 4
    else:
        r = e.deflate([data], level=6)[0]
        assert r.data == O.deflate(data, 6)
    by_src = collections.Counter()
    for k, c in cnt.items():
        j = bisect.bisect_right(loc_lines, mg.ins[k].line) - 1
        by_src[(locs[j][2], locs[j][1]) if j >= 0 else (0, 0)] += c
    tot = sum(cnt.values())
    src = open(os.path.join(ROOT, "sharpziplib_amd", "csrc", unit + ".hip")).read().splitlines()
    main_ids = [k for k, v in files.items() if v.endswith(unit + ".hip")]
    print("%s: %d wave-instructions, %.2f per byte of the stream" % (ksel, tot, tot / nbytes))
    acc = 0
    for (fid, ln), c in by_src.most_common(70):
        acc += c
        name = files.get(fid, "?").split("/")[-1]
        txt = src[ln - 1].strip()[:120] if fid in main_ids and 0 < ln <= len(src) else ("(no line: prologue, address arithmetic, spills)" if ln == 0 else "")
        print("%5.1f%% %5.1f%%  %s:%d  %s" % (100.0 * c / tot, 100.0 * acc / tot, name, ln, txt))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
