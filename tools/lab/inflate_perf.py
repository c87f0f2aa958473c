"""Inflate timings: one big member (bit-serial per stream) and many members."""
import sys, ctypes, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib()
eng = Engine()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
msz = int(sys.argv[3]) if len(sys.argv) > 3 else (1 << 20)
for kind in ('enwik', 'logs'):
    d = C.generate(kind, 0xE9, 0, mb << 20)
    comp = eng.deflate([d], level=6)[0].data
    for rep in range(2):
        t = time.time(); (r, cons), = eng.inflate([comp], [d.size]); dt = time.time() - t
        km = eng.timing()['inflate_ms']
    print(f"single {kind} {mb} MiB: kernel {km:.1f} ms -> {mb/(km/1e3):.1f} MiB/s out ({len(comp)/2**20/(km/1e3):.1f} MiB/s in) ok={r.data == d.tobytes()} wall={dt*1e3:.0f}ms", flush=True)
d = C.generate('enwik', 0xE9, 0, nm * msz)
parts = [d[i * msz:(i + 1) * msz] for i in range(nm)]
res = eng.deflate(parts, level=6)
comps = [r.data for r in res]
for rep in range(2):
    t = time.time(); out = eng.inflate(comps, [msz] * nm); dt = time.time() - t
    km = eng.timing()['inflate_ms']
ok = all(o[0].data == p.tobytes() for o, p in zip(out, parts))
print(f"{nm} x {msz>>10} KiB members: kernel {km:.1f} ms -> {nm*msz/2**20/(km/1e3):.0f} MiB/s out ok={ok} wall={dt*1e3:.0f}ms", flush=True)
