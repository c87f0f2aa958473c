import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from sharpziplib_amd import corpus, _lib
from sharpziplib_amd.deflater import Deflater
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 512) << 20
data = corpus.generate("enwik", 0xE9, 0, n)
buf = np.zeros(64 << 20, np.uint8)
for rep in range(2):
    _lib.lib().szl_debug_set(b"SZL_DEBUG", rep)
    d = Deflater(6, True)
    t0 = time.perf_counter()
    ts = []
    for o in range(0, n, 16 << 20):
        ta = time.perf_counter(); d.SetInput(data[o:o + (16 << 20)]); tb = time.perf_counter(); d.Deflate(buf); ts.append((tb - ta) * 1e3); ts.append((time.perf_counter() - tb) * 1e3)
    t1 = time.perf_counter()
    d.Finish()
    tot = 0
    while not d.IsFinished:
        tot += d.Deflate(buf)
    t2 = time.perf_counter()
    if rep: print('SetInput / Deflate ms:', ' '.join('%.2f' % x for x in ts[:24]), '...', ' '.join('%.2f' % x for x in ts[-8:]))
    if rep: print('python: t0 %.2f t1 %.2f t2 %.2f (time.monotonic ms)' % (t0 * 1e3, t1 * 1e3, t2 * 1e3))
    print("writes %.1f ms finish %.1f ms out %d parts %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, tot, _lib.lib().szl_deflater_debug_pipe_parts(d._h)), flush=True)
    del d
