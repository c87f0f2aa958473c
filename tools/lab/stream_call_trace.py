"""One 64 KiB entry per Reset / SetInput / Finish / Deflate through the streaming Deflater under `rocprofv3 --kernel-trace` (the unchanged
ZipOutputStream path as the device sees it); tools/lab/small_call_trace_report.py reads the trace.  Also prints the host's wall time per step."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C
from sharpziplib_amd.deflater import Deflater
kib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = kib << 10
data = C.generate('enwik', 5, 0, n * 40)
d = Deflater(6, True)
buf = np.zeros(n + 4096, np.uint8)
lap = [0.0] * 4
for i in range(40):
    t0 = time.perf_counter(); d.Reset()
    t1 = time.perf_counter(); d.SetInput(data[i * n:(i + 1) * n])
    t2 = time.perf_counter(); d.Finish()
    while not d.IsFinished:
        d.Deflate(buf)
    t3 = time.perf_counter()
    if i >= 20:
        lap[0] += t1 - t0; lap[1] += t2 - t1; lap[2] += t3 - t2
    time.sleep(0.002)
print("per entry (us): Reset %.1f  SetInput %.1f  Finish+Deflate %.1f" % tuple(x / 20 * 1e6 for x in lap[:3]))
