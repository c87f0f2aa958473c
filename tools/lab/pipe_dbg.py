import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from sharpziplib_amd import corpus, _lib
from sharpziplib_amd.deflater import Deflater
n = 512 << 20
data = corpus.generate("enwik", 0xE9, 0, n)
buf = np.zeros(64 << 20, np.uint8)
for rep in range(2):
    _lib.lib().szl_debug_set(b"SZL_DEBUG", rep)
    d = Deflater(6, True)
    t0 = time.perf_counter()
    for o in range(0, n, 16 << 20):
        d.SetInput(data[o:o + (16 << 20)]); d.Deflate(buf)
    t1 = time.perf_counter()
    d.Finish()
    tot = 0
    while not d.IsFinished:
        tot += d.Deflate(buf)
    t2 = time.perf_counter()
    print("writes %.1f ms finish %.1f ms out %d parts %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, tot, _lib.lib().szl_deflater_debug_pipe_parts(d._h)), flush=True)
    del d
