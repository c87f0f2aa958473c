"""One small stream per call (the per-entry cost of the unchanged ZipOutputStream path): stage times and wall time per call."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for kib in (4, 64, 256, 1024):
    d = C.generate('enwik', 7, 0, kib << 10)
    for _ in range(5):
        eng.deflate([d], level=6)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = eng.deflate([d], level=6)
    wall = (time.perf_counter() - t0) / reps * 1e3
    tm = eng.timing()
    print(f"{kib:5d} KiB: wall {wall:.3f} ms/call; device {tm['total_ms']:.3f} ms = A {tm['links_ms']:.3f} pilot {tm['pilot_ms']:.3f} B {tm['match_ms']:.3f} C {tm['parse_ms']:.3f} D {tm['blocks_ms']:.3f} E {tm['encode_ms']:.3f} ck {tm['checksum_ms']:.3f}", flush=True)
