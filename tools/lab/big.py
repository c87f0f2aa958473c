"""64-bit paths: one stream larger than 2^31 bytes (config 5 shape, reduced): L9 logs 2.5 GiB, round trip through zlib."""
import sys, ctypes, time, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else (5 << 29)
level = int(sys.argv[2]) if len(sys.argv) > 2 else 9
t = time.time(); data = C.generate('logs', 0x106, 0, n); print(f'gen {n/2**30:.2f} GiB in {time.time()-t:.1f}s', flush=True)
arr, in_total, out_total = Engine.layout([n])
hout = np.zeros(out_total + 8, np.uint8)
t = time.time()
_lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, 1, level, 0, _lib.F_NOWRAP | _lib.F_CRC32), 'batch')
dt = time.time() - t
tm = eng.timing()
print(f"L{level} logs {n/2**30:.2f} GiB wall={dt:.2f}s gpu={tm['total_ms']:.0f}ms -> {n/2**20/(tm['total_ms']/1e3):.0f} MiB/s A {tm['links_ms']:.0f} B {tm['match_ms']:.0f} C {tm['parse_ms']:.0f} D {tm['blocks_ms']:.0f} E {tm['encode_ms']:.0f} ratio={tm['out_bytes']/tm['in_bytes']:.4f} tokens={tm['tokens']} blocks={tm['blocks']} unmerged={tm['ranges_unmerged']} fb={tm['fallback_walks']}", flush=True)
comp = hout[:arr[0].out_len].tobytes()
t = time.time()
do = zlib.decompressobj(-15)
ok = True; pos = 0
for i in range(0, len(comp), 1 << 24):
    chunk = do.decompress(comp[i:i + (1 << 24)])
    if chunk != data[pos:pos + len(chunk)].tobytes():
        ok = False; break
    pos += len(chunk)
chunk = do.flush(); ok = ok and chunk == data[pos:pos + len(chunk)].tobytes(); pos += len(chunk)
print('zlib roundtrip', ok and pos == n, f'({time.time()-t:.1f}s)', 'crc', arr[0].crc32 == zlib.crc32(data), flush=True)
