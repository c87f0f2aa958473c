"""Exercise the other BASELINE configs at reduced/full size on the GPU box (timings + correctness spot checks)."""
import sys, ctypes, time, zlib, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib()
eng = Engine()
which = sys.argv[1:] or ['c3', 'c4', 'c5']

if 'c3' in which:   # many small streams: N x 64 KiB
    N = int(os.environ.get('SZL_C3_N', '20000'))
    t = time.time(); data = C.generate('dickens', 0x21B0, 0, N * 65536); print(f'gen {time.time()-t:.1f}s', flush=True)
    arr, in_total, out_total = Engine.layout([65536] * N)
    hout = np.zeros(out_total + 8, np.uint8)
    for rep in range(2):
        t = time.time()
        _lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, N, 6, 0, _lib.F_NOWRAP | _lib.F_CRC32), 'batch')
        dt = time.time() - t
        tm = eng.timing()
        print(f"c3: {N} x 64KiB wall={dt*1e3:.0f}ms gpu={tm['total_ms']:.1f}ms -> {N*65536/2**20/(tm['total_ms']/1e3):.0f} MiB/s stages ck {tm['checksum_ms']:.1f} A {tm['links_ms']:.1f} B {tm['match_ms']:.1f} C {tm['parse_ms']:.1f} D {tm['blocks_ms']:.1f} E {tm['encode_ms']:.1f} ratio={tm['out_bytes']/tm['in_bytes']:.4f}", flush=True)
    for i in (0, 1, N // 2, N - 1):
        s = arr[i]
        comp = hout[s.out_off:s.out_off + s.out_len].tobytes()
        d = data[s.in_off:s.in_off + s.in_len]
        assert comp == O.deflate(d, 6), i
        assert s.crc32 == zlib.crc32(d.tobytes())
    print('c3 spot checks ok', flush=True)
    # inflate them all back on the device
    bufs_total = sum(int(s.out_len) for s in arr)
    istreams = (_lib.Stream * N)()
    io = 0
    cin = np.zeros(bufs_total + 4 * N + 8, np.uint8)
    for i in range(N):
        s = arr[i]
        cin[io:io + s.out_len] = hout[s.out_off:s.out_off + s.out_len]
        istreams[i].in_off, istreams[i].in_len, istreams[i].out_off, istreams[i].out_cap = io, s.out_len, i * 65536, 65536
        io += (int(s.out_len) + 3) & ~3
    dout = np.zeros(N * 65536 + 8, np.uint8)
    for rep in range(2):
        t = time.time()
        _lib.check(L.szl_inflate_batch_host(eng._h, cin.ctypes.data, dout.ctypes.data, istreams, N, _lib.F_NOWRAP), 'inflate')
        dt = time.time() - t
        km = eng.timing()['inflate_ms']
        print(f"c3 inflate: wall={dt*1e3:.0f}ms kernel={km:.1f}ms -> {N*65536/2**20/(km/1e3):.0f} MiB/s (device)", flush=True)
    assert all(s.status == 0 for s in istreams)
    assert np.array_equal(dout[:N * 65536], data)
    print('c3 inflate roundtrip ok', flush=True)

if 'c4' in which:   # multi-member inflate: 4 MiB members
    M, msz = int(sys.argv[2]) if len(sys.argv) > 2 else 128, 4 << 20
    data = C.generate('enwik', 0xEA, 0, M * msz)
    arr, in_total, out_total = Engine.layout([msz] * M)
    hout = np.zeros(out_total + 8, np.uint8)
    _lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, M, 6, 0, _lib.F_NOWRAP), 'batch')
    tm = eng.timing(); print(f"c4 prep deflate {M} x 4MiB gpu={tm['total_ms']:.1f}ms", flush=True)
    istreams = (_lib.Stream * M)()
    io = 0
    cin = np.zeros(sum(int(s.out_len) for s in arr) + 4 * M + 8, np.uint8)
    for i in range(M):
        s = arr[i]
        cin[io:io + s.out_len] = hout[s.out_off:s.out_off + s.out_len]
        istreams[i].in_off, istreams[i].in_len, istreams[i].out_off, istreams[i].out_cap = io, s.out_len, i * msz, msz
        io += (int(s.out_len) + 3) & ~3
    dout = np.zeros(M * msz + 8, np.uint8)
    for rep in range(2):
        t = time.time()
        _lib.check(L.szl_inflate_batch_host(eng._h, cin.ctypes.data, dout.ctypes.data, istreams, M, _lib.F_NOWRAP | _lib.F_CRC32), 'inflate')
        dt = time.time() - t
        km = eng.timing()['inflate_ms']
        print(f"c4 inflate {M} x 4MiB members: wall={dt*1e3:.0f}ms kernel={km:.1f}ms -> {M*msz/2**20/(km/1e3):.0f} MiB/s out (device), {M*msz/2**20/dt:.0f} MiB/s wall", flush=True)
    assert all(s.status == 0 for s in istreams) and np.array_equal(dout[:M * msz], data)
    print('c4 roundtrip ok', flush=True)

if 'c5' in which:   # L9 on repetitive logs
    n = 256 << 20
    data = C.generate('logs', 0x106, 0, n)
    arr, in_total, out_total = Engine.layout([n])
    hout = np.zeros(out_total + 8, np.uint8)
    for rep in range(2):
        _lib.check(L.szl_deflate_batch_host(eng._h, data.ctypes.data, hout.ctypes.data, arr, 1, 9, 0, _lib.F_NOWRAP), 'batch')
        tm = eng.timing()
        print(f"c5: L9 logs {n>>20}MiB gpu={tm['total_ms']:.1f}ms -> {n/2**20/(tm['total_ms']/1e3):.0f} MiB/s A {tm['links_ms']:.1f} B {tm['match_ms']:.1f} C {tm['parse_ms']:.1f} D {tm['blocks_ms']:.1f} E {tm['encode_ms']:.1f} ratio={tm['out_bytes']/tm['in_bytes']:.4f} unmerged={tm['ranges_unmerged']} fb={tm['fallback_walks']}", flush=True)
    comp = hout[:arr[0].out_len].tobytes()
    assert zlib.decompress(comp, -15) == data.tobytes()
    t = time.time(); ref = O.deflate(data[:64 << 20], 9); print(f'oracle L9 64MiB: {64/(time.time()-t):.1f} MiB/s', flush=True)
    r = eng.deflate([data[:64 << 20]], level=9)[0].data
    print('c5 64MiB oracle eq', r == ref, flush=True)
