import sys, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
eng = Engine()
data = C.four_symbol(250000)
for lv in (0, 1, 6, 9):
    comp = O.deflate(data, lv)
    for rep in range(3):
        (r, consumed), = eng.inflate([comp], [data.size])
        ok = r.data == data.tobytes()
        bad = -1
        if not ok and len(r.data) == data.size:
            a = np.frombuffer(r.data, np.uint8); bad = int(np.nonzero(a != data)[0][0])
        print('level', lv, 'rep', rep, 'status', r.status, 'len', len(r.data), 'ok', ok, 'consumed', consumed, len(comp), 'first bad', bad, flush=True)
