"""One buffer the small-call soak found (level 7, Filtered, 192336 bytes: ONE range of 64 positions never merges): where do the device's
tokens leave the oracle's?   python tools/lab/small_call_mismatch_dbg.py file.npy level strategy [lab]   (knobs through the environment)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path: sys.path.insert(0, p)
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import _lib
if len(sys.argv) > 4 and sys.argv[4] == "lab": _lib._lib = _lib.lab_lib()
from sharpziplib_amd.batch import Engine
b = np.load(sys.argv[1]); level = int(sys.argv[2]); strategy = int(sys.argv[3])
knobs = {k: v for k, v in os.environ.items() if k.startswith("SZL_")}
eng = Engine()
want, tr = O.deflate(b, level=level, nowrap=True, strategy=strategy, trace=True)
got = eng.deflate([b], level=level, strategy=strategy, nowrap=True)[0]
tm = eng.timing()
_, _, _, tok = eng.debug_fetch(b.size)
wt = tr["tokens"]; m = min(tok.size, wt.size)
wl = np.where(wt >> 16, wt & 0xFFFF, 1).astype(np.int64); wpos = np.concatenate([[0], np.cumsum(wl)])
d = np.flatnonzero(tok[:m] != wt[:m])
print("%s %s: bytes %s (%d against %d), tokens %d against %d, unmerged ranges %d, fallback walks %d"
      % ("lab" if len(sys.argv) > 4 else "product", knobs, "EQUAL" if got.data == want else "DIFFERENT", len(got.data), len(want), tok.size, wt.size,
         tm["ranges_unmerged"], tm["fallback_walks"]), flush=True)
if d.size:
    i = int(d[0]); p = int(wpos[i])
    print("   first different token %d at input position %d (range of 64: %d, offset %d): device %s | oracle %s"
          % (i, p, p // 64, p % 64, [hex(int(x)) for x in tok[i:i + 6]], [hex(int(x)) for x in wt[i:i + 6]]))
    gl = np.where(tok >> 16, tok & 0xFFFF, 1).astype(np.int64)
    print("   the device's tokens cover %d bytes, the oracle's %d; token lengths agree again from token %s"
          % (int(gl.sum()), int(wl.sum()), "-" if tok.size != wt.size else int(d[-1]) + 1))
