"""stage times of one stream of a data class (python tools/gpu_class_stats.py class MiB level)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
name, mb, lv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = mb << 20
d = {'zeros': lambda: C.zeros(n), 'period10': lambda: C.period10(n), 'mixed': lambda: C.mixed(n, seed=5),
     'bytes256': lambda: np.resize(np.frombuffer(bytes(range(256)) + b"xyz", np.uint8), n)}.get(name, lambda: C.generate(name, 7, 0, n))()
for rep in range(2):
    r = eng.deflate([d], level=lv)[0]
print(name, mb, lv, eng.timing())
