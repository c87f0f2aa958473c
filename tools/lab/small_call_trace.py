"""One 64 KiB stream per call under `rocprofv3 --kernel-trace`: which kernels a call launches, in order, how long each runs and how long the
device idles between them (the per-entry cost of the unchanged ZipOutputStream path).  Prints nothing itself;
tools/lab/small_call_trace_report.py reads the trace."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
kib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = C.generate('enwik', 7, 0, kib << 10)
for _ in range(12):
    eng.deflate([d], level=6)
