"""Stage B forms (full / on demand, single walkers or walker+scout pairs) on different data and levels: timings + parity."""
import sys, os, subprocess
cases = [('logs', 9, 256), ('logs', 6, 512), ('enwik', 6, 512)]
code = r'''
import sys, ctypes, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
kind, lv, mb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
d = C.generate(kind, 0x106, 0, mb << 20)
for rep in range(2):
    r = eng.deflate([d], level=lv)[0]
tm = eng.timing()
small = d[:8 << 20]
ok = eng.deflate([small], level=lv)[0].data == O.deflate(small, lv)
print(f"{kind} L{lv} {mb}MiB total={tm['total_ms']:.1f} B={tm['match_ms']:.1f} C={tm['parse_ms']:.1f} -> {mb/(tm['total_ms']/1e3):.0f} MiB/s eq8MiB={ok}", flush=True)
'''
open('/tmp/_lazy_case.py', 'w').write(code)
envs = [{'SZL_MATCH_MODE': '0'}, {'SZL_MATCH_MODE': '1', 'SZL_STRIDE': '16'}, {'SZL_MATCH_MODE': '1', 'SZL_STRIDE': '16', 'SZL_PAIR': '1'},
        {'SZL_MATCH_MODE': '1', 'SZL_STRIDE': '32', 'SZL_PAIR': '1'}]
for kind, lv, mb in cases:
    for env in envs:
        e = dict(os.environ); e.update(env); e['SZL_DEBUG'] = '1'
        out = subprocess.run([sys.executable, '/tmp/_lazy_case.py', kind, str(lv), str(mb)], env=e, capture_output=True, text=True)
        stage = [l for l in out.stderr.splitlines() if 'stage B' in l]
        print(env, out.stdout.strip(), '|', stage[-2][12:80] if len(stage) >= 2 else out.stderr[-200:], flush=True)
