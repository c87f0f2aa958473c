"""What the library keeps when its objects are gone (CPU, through tools/gfxsim's fake HIP runtime, which counts live hipMalloc /
hipHostMalloc bytes).  The pools exist for the caller who makes one Deflater after another (S/GZip/GzipOutputStream.cs:87); a host
that is done should get its memory back: down to SZL_IDLE_KEEP_MIB when the last streaming object is destroyed, all of it on
szl_trim() (include/szl.h)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes, os, sys
ROOT = %r
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from gfxsim import harness
rt = harness.use(fast_probe=True)
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import _lib, corpus as C
from sharpziplib_amd.deflater import Deflater
from sharpziplib_amd.inflater import Inflater
L = _lib.lib()
fake = ctypes.CDLL(os.path.join(harness.BUILD, "libfakehip.so"), mode=ctypes.RTLD_GLOBAL)      # (the one the harness loaded)
fake.fakehip_live_device_bytes.restype = ctypes.c_size_t
fake.fakehip_live_host_bytes.restype = ctypes.c_size_t
def live():
    return int(fake.fakehip_live_device_bytes()), int(fake.fakehip_live_host_bytes())
def one_round():
    d = Deflater(6, True)
    d.SetInput(data); d.Finish()
    got = bytearray()
    while not d.IsFinished:
        n = d.Deflate(buf); got += buf[:n].tobytes()
    assert bytes(got) == O.deflate(data, 6)
    held = live()
    i = Inflater(True)
    i.SetInput(bytes(got))
    back = np.zeros(40000, np.uint8)
    n = i.Inflate(back)
    assert back[:n].tobytes() == data.tobytes()
    del i
    del d                                                 # (__del__ -> szl_*_destroy)
    return held
data = C.generate("enwik", 5, 0, 6000)
buf = np.zeros(1 << 16, np.uint8)
one_round()                                               # (per-process tables the first call leaves for good: probes, constant tables)
L.szl_trim()
base = live()
L.szl_debug_set(b"SZL_IDLE_KEEP_MIB", 0)                 # the idle state keeps nothing ...
L.szl_debug_set(b"SZL_IDLE_TRIM_MS", 0)                  # ... and begins at once (default: after two seconds without an object)
for k in range(2):                                        # one object after another
    held = one_round()
    after = live()
    print("round", k, "held", held, "after the last object", after)
    assert held[0] > base[0] + 50000                     # work space on the device while the objects live
    assert after[0] <= base[0] and after[1] <= base[1] + 4096, (base, after)   # (an idle engine object keeps its 256-byte read-back page)
L.szl_debug_set(b"SZL_IDLE_KEEP_MIB", -2147483648)
one_round()
kept = live()
assert kept[0] > base[0]                                  # the default keeps an idle engine's (small) work space: the next object is fast
assert L.szl_trim() == 0
assert live()[0] <= base[0] and live()[1] <= base[1], (base, live())
print("ok: default idle state kept", kept, "szl_trim ->", live())
'''


def test_pools_shrink_with_the_last_object_and_trim_frees_all():
    env = dict(os.environ, GFXSIM_LAB="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ok: default idle state kept" in r.stdout
