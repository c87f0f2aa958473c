"""Round-5 A/B of the variants round 4 built without GPU minutes (all behind knobs, bytes checked): inflate slots, pinned hand-out, stage-B thresholds.

    python tools/gpu_lab.py r5_ab [--mib 512] [--quick]

Everything runs in one process on one device; every line is `what | knob = value | time` and every output is compared with the input /
the default's bytes.  Nothing here changes a default: the winners are set in the source afterwards.
  1. the chunk-parallel Inflater on one member and on 64 x 1 MiB members: SZL_INF_SLOTS_PER_CU = 8 / 10 (chunk sizing) x SZL_INF_DENSE = 0 / 1
     (k_inflate<true,2,DENSE>: 168 registers) x SZL_INF_TRIM_TAIL = 0 / 1 (no tail round of stragglers: 64 x 4 MiB members are 2112 jobs for
     2048 slots with the default sizing) — round 4's last GPU seconds: slots 10 + DENSE, one member -15 %, the members 3 x SLOWER (jobs that span two of the
     smaller chunks overran their staging regions and were run again, a pass each: SZL_INF_REG_BY_SPAN = 1 sizes a job's region by its span)
  2. InflaterInputStream over that member with 16 MiB and 64 MiB buffers: SZL_INF_PINNED = 0 (default) / 1
  3. raw deflate level 6 of the same text (the bench step): the default, then the stage-B knobs one at a time (SZL9_FTH, SZL_TILE_LEN);
     with --lab also SZL_SPEC_WB = 0 / 1 (k_spec_win's write-back, four ranges per store: -14 % of the kernel's instructions on the interpreter)
"""
import argparse
import hashlib
import io
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                    # noqa: E402

from sharpziplib_amd import _lib, corpus                # noqa: E402
from sharpziplib_amd.batch import Engine                # noqa: E402
from sharpziplib_amd.inflater import Inflater           # noqa: E402
from sharpziplib_amd.streams import InflaterInputStream  # noqa: E402

FORGET = -2147483648
ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=512)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--lab", action="store_true", help="the laboratory library (libszl_amd_lab.so): also SZL_SPEC_WB, the four-ranges-per-store write-back of k_spec_win")
ap.add_argument("--kib", type=int, default=0, help="(tools/gfxsim dry run) a member of this many KiB instead of --mib")
a = ap.parse_args()
if a.lab:
    _lib._lib = _lib.lab_lib()
L = _lib.lib()


def knob(name, v):
    L.szl_debug_set(name.encode(), int(v))


n = (a.kib << 10) if a.kib else (a.mib << 20)
if a.kib:
    a.mib = a.kib / 1024.0
plain = corpus.generate("enwik", 0xE9, 0, n)
eng = Engine()
r0 = eng.deflate([plain], level=6, crc32=True)[0]
comp = r0.data
want = hashlib.sha256(plain.tobytes()).hexdigest()
print("member: %.2f MiB of text -> %d bytes" % (a.mib, len(comp)), flush=True)

# ---- 1. the symbol pass: chunk sizing (jobs per CU the chunks are cut for) x register budget of the kernel, on two shapes
msz = (1 << 20) if not a.kib else 256            # (64 x 1 MiB: the shape that stands at 3.4 GiB/s, with a second symbol pass over 35 jobs)
parts = [plain[i * msz:(i + 1) * msz] for i in range(min(64, n // msz))]
mcomps = [r.data for r in eng.deflate(parts, level=6)] if parts else []
for slots, dense, trim, span in ((8, 0, 0, 0), (8, 0, 1, 0), (8, 0, 0, 1), (8, 0, 1, 1), (10, 0, 0, 1), (10, 1, 0, 0), (10, 1, 0, 1), (10, 1, 1, 1), (8, 0, 0, 0)):
    knob("SZL_INF_SLOTS_PER_CU", slots); knob("SZL_INF_DENSE", dense); knob("SZL_INF_TRIM_TAIL", trim); knob("SZL_INF_REG_BY_SPAN", span)
    best = 1e9
    for rep in range(3):
        (r, used), = eng.inflate([comp], [n], crc32=True)
        assert r.status == 0 and used == len(comp) and r.crc32 == r0.crc32
        best = min(best, eng.timing()["inflate_ms"])
    assert hashlib.sha256(r.data).hexdigest() == want
    bm = 1e9
    for rep in range(3 if mcomps else 0):
        out = eng.inflate(mcomps, [msz] * len(mcomps))
        bm = min(bm, eng.timing()["inflate_ms"])
    assert all(o[0].data == p.tobytes() for o, p in zip(out, parts)) if mcomps else True
    print("inflate | SZL_INF_SLOTS_PER_CU = %2d SZL_INF_DENSE = %d SZL_INF_TRIM_TAIL = %d SZL_INF_REG_BY_SPAN = %d | one member %8.2f ms (%5.1f GiB/s) | %d x 1 MiB members %8.2f ms" % (
        slots, dense, trim, span, best, a.mib / 1024 / (best * 1e-3), len(mcomps), bm if mcomps else 0.0), flush=True)
knob("SZL_INF_SLOTS_PER_CU", FORGET); knob("SZL_INF_DENSE", FORGET); knob("SZL_INF_TRIM_TAIL", FORGET); knob("SZL_INF_REG_BY_SPAN", FORGET)

# ---- 1b. many mid-size members: one wavefront per member (the library's choice above 128 streams) against the chunked form, with and
#          without the span-sized regions (512 x 1 MiB stood at 51.8 ms = 9.65 GiB/s one wavefront each; chunked it lost while every member cost a second pass)
many = [plain[i * msz:(i + 1) * msz] for i in range(min(512, n // msz))]
if len(many) > 128:
    mc = [r.data for r in eng.deflate(many, level=6)]
    for parmin, span in ((FORGET, 0), (128, 0), (128, 1), (FORGET, 0)):
        knob("SZL_INF_PAR_MIN_KIB", parmin); knob("SZL_INF_REG_BY_SPAN", span)
        bm = 1e9
        for rep in range(3):
            out = eng.inflate(mc, [msz] * len(mc))
            bm = min(bm, eng.timing()["inflate_ms"])
        assert all(o[0].data == p.tobytes() for o, p in zip(out, many))
        print("inflate %d x 1 MiB members | SZL_INF_PAR_MIN_KIB = %s SZL_INF_REG_BY_SPAN = %d | %8.2f ms (%5.1f GiB/s), %d chunk jobs" % (
            len(mc), "default" if parmin == FORGET else parmin, span, bm, len(mc) / 1024 / (bm * 1e-3), L.szl_engine_debug_par_jobs(eng._h)), flush=True)
    knob("SZL_INF_PAR_MIN_KIB", FORGET); knob("SZL_INF_REG_BY_SPAN", FORGET)

# ---- 2. the unchanged-host read path with room
for bufsz in ((64 << 20,) if a.quick else (16 << 20, 64 << 20)):
    for pinned in (0, 1, 0, 1):
        knob("SZL_INF_PINNED", pinned)
        best = 0.0
        for rep in range(2):
            inf = Inflater(True)
            st = InflaterInputStream(io.BytesIO(comp), inf, bufsz)
            out = np.zeros(4 << 20, np.uint8)
            h = hashlib.sha256() if rep == 0 else None
            t0 = time.perf_counter()
            got = 0
            while True:
                k = st.Read(out, 0, out.size)
                if k <= 0:
                    break
                got += k
                if h:
                    h.update(out[:k].tobytes())
            dt = time.perf_counter() - t0
            assert got == n and inf.RemainingInput == 0 and inf.TotalIn == len(comp)
            if h:
                assert h.hexdigest() == want
            else:
                best = a.mib / dt
        print("InflaterInputStream %2d MiB buffers | SZL_INF_PINNED = %d | %8.1f MiB/s (%d pieces through the chunk-parallel decoder)" % (
            bufsz >> 20, pinned, best, L.szl_inflater_debug_bulk_calls(inf._h)), flush=True)
knob("SZL_INF_PINNED", FORGET)

# ---- 3. the bench step and the stage-B knobs
ref = hashlib.sha256(comp).hexdigest()


def step(label):
    best = (1e9, None)
    for rep in range(3):
        r = eng.deflate([plain], level=6)[0]
        t = eng.timing()
        if t["total_ms"] < best[0]:
            best = (t["total_ms"], t)
    assert hashlib.sha256(r.data).hexdigest() == ref, label
    t = best[1]
    print("deflate level 6 | %-24s | total %7.2f ms  A %5.2f  B %6.2f  C %5.2f  D %5.2f  E %5.2f  -> %6.1f MiB/s" % (
        label, t["total_ms"], t["links_ms"], t["match_ms"], t["parse_ms"], t["blocks_ms"], t["encode_ms"], a.mib / (t["total_ms"] * 1e-3)), flush=True)


step("defaults")
if not a.quick:
    for name, values in ((("SZL_SPEC_WB", (0, 1, 0, 1)),) if a.lab else ()) + (("SZL9_FTH", (12, 16, 20)), ("SZL_TILE_LEN", (18432, 20480, 21504))):
        for v in values:
            knob(name, v)
            step("%s = %d" % (name, v))
        knob(name, FORGET)
step("defaults (again)")
eng.close()
