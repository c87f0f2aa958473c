"""The product's MACHINE CODE on the CPU: tools/gfxsim interprets the gfx950 assembly of every kernel (hipcc -S of the same
sources, same flags) under the product's own host objects linked against a fake HIP runtime, and the results are compared with
the oracle exactly as the GPU tests do (levels 0-9, data classes, the streaming objects with SetLevel / Reset / dictionary,
Inflater on valid, corrupted and quirk-set streams, one member on many wavefronts, the window pipeline and stage A / B's other forms, two devices behind the multi-device entry points).  Each suite of tools/gfxsim/suite.py runs in its own process (the
interpreter replaces the package's library handle; the other CPU tests must keep seeing the real one), all of them at once.

This is evidence about the instruction text, not about the chip: no timing, no inter-wavefront memory ordering, and the LDS
exchange order of DESIGN 4.1 is modelled, not proved.  The GPU tests (-m gpu) stay the parity tests proper.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["inflate_parallel", "inflate_stream_bulk", "inflate_corrupt", "forms", "lab_forms@lab", "multi_device", "deflate_levels", "deflater_object", "framing", "deflate_shapes", "inflate",
          "exchange_order@desc", "exchange_order@flaky:30:130"]   # longest first; @ = the lane order ds_wrxchg is served in (fault injection) / the laboratory library
# (`inflate_dense` — the 3-wavefronts-per-SIMD build of the symbol pass, measured and not adopted in round 5 — lives in the laboratory library and is
# run by hand: GFXSIM_LAB=1 python tools/gfxsim/suite.py inflate_dense)
HIPCC = "/opt/rocm/bin/hipcc"
# ≈720 CPU-seconds in all, spread over the cores (140 s of wall time on 8); a box with fewer than 4 cores runs the core suites only
if (os.cpu_count() or 1) < 4:
    SUITES = [s for s in SUITES if s in ("deflate_levels", "deflater_object", "deflate_shapes", "inflate", "inflate_corrupt", "framing", "exchange_order@desc")]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc is needed for the device assembly")


_PROCS = None


def start_all():
    """start every suite's process (idempotent).  tests/conftest.py calls this as soon as collection shows that this module will run, so
    that the interpreter works on the other cores while the rest of the CPU suite runs on one."""
    global _PROCS
    if _PROCS is None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from gfxsim import harness
        harness.build()                               # libszl_amd.so's objects, the fake runtime, the kernels' assembly (cached)
        if any(s.endswith("@lab") for s in SUITES):
            subprocess.check_call([sys.executable, "-c", "import os,sys; os.environ['GFXSIM_LAB']='1'; sys.path.insert(0, %r); from gfxsim import harness; harness.build()" % os.path.join(ROOT, "tools")])
        env = dict(os.environ, PYTHONHASHSEED="0", GFXSIM_POISON="1")   # LDS and fresh device memory start as garbage, as on the device
        env.pop("SZL_DEBUG", None)
        _PROCS = {s: subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gfxsim", "suite.py"), s.split("@")[0]], cwd=ROOT,
                                      env=dict(env, **({("GFXSIM_LAB" if s.endswith("@lab") else "GFXSIM_XCHG"): s.split("@")[1]} if "@" in s else {})),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for s in SUITES}
    return _PROCS


def stop_all():
    for p in (_PROCS or {}).values():
        if p.poll() is None:
            p.kill()


@pytest.fixture(scope="module")
def runs():
    yield start_all()
    stop_all()


@pytest.mark.parametrize("suite", SUITES)
def test_machine_code_equals_oracle(runs, suite):
    p = runs[suite]
    try:
        out, _ = p.communicate(timeout=1500)
    except subprocess.TimeoutExpired:
        p.kill()
        pytest.fail("suite %s did not finish" % suite)
    assert p.returncode == 0 and ("ok %s:" % suite.split("@")[0]) in out, out[-4000:]
    print(out.strip().splitlines()[-1])
