"""full search vs on-demand stage B across data classes (is there any data where the on-demand form still wins?)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = mb << 20
classes = {
    'zeros': lambda: C.zeros(n), 'period10': lambda: C.period10(n), 'four_symbol': lambda: C.four_symbol(n),
    'logs': lambda: C.generate('logs', 7, 0, n), 'dickens': lambda: C.generate('dickens', 7, 0, n), 'enwik': lambda: C.generate('enwik', 7, 0, n),
    'random': lambda: C.random_bytes(n, seed=3), 'mixed': lambda: C.mixed(n, seed=5),
    'bytes256': lambda: np.resize(np.frombuffer(bytes(range(256)) + b"xyz", np.uint8), n),
    'logs_x4': lambda: np.tile(C.generate('logs', 9, 0, n // 4), 4),
}
for name, mk in classes.items():
    d = mk()
    for lv in (6, 9):
        row = []
        ref = None
        for mode in (0, 1):
            eng.debug_match_mode(mode)
            for rep in range(2):
                r = eng.deflate([d], level=lv)[0]
            tm = eng.timing()
            used = eng.debug_match_mode()
            if ref is None: ref = r.data
            row.append(f"{'od' if used else 'full'} {tm['total_ms']:.1f} (B {tm['match_ms']:.1f}){'' if r.data == ref else ' DIFF'}")
        print(f"{name:12s} L{lv}: " + "   ".join(row), flush=True)
eng.debug_match_mode(-1)
