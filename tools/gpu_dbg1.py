import sys, ctypes, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_ffi as O
from sharpziplib_amd import corpus as C, _lib
L = _lib.lib()
print('devices', L.szl_device_count())
eng = L.szl_engine_create()
assert eng, L.szl_last_error()
def run(name, data, level=6, flags=_lib.F_NOWRAP|_lib.F_CRC32|_lib.F_ADLER32):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = data.size
    cap = int(L.szl_deflate_bound(n)) + 16
    cap = (cap + 3) & ~3
    out = np.zeros(cap, dtype=np.uint8)
    st = _lib.Stream(0, n, 0, cap, 0, 0, 0, 0, 0, 0)
    inb = np.concatenate([data, np.zeros(8, np.uint8)])
    t=time.time()
    rc = L.szl_deflate_batch_host(eng, inb.ctypes.data, out.ctypes.data, ctypes.byref(st), 1, level, 0, flags)
    dt=time.time()-t
    if rc != 0:
        print(name, 'FAILED rc', rc, L.szl_last_error()); return False
    tm = _lib.Timing(); L.szl_engine_last_timing(eng, ctypes.byref(tm))
    comp = out[:st.out_len].tobytes()
    ref, tr = O.deflate(data, level, nowrap=True, trace=True)
    ok = comp == ref
    okc = st.crc32 == O.crc32(data) and st.adler32 == O.adler32(data)
    print(f"{name:10s} L{level} n={n} out={st.out_len} ref={len(ref)} bytes_eq={ok} cksum_eq={okc} wall={dt*1e3:.1f}ms gpu={tm.total_ms:.2f}ms [ck {tm.checksum_ms:.2f} A {tm.links_ms:.2f} B {tm.match_ms:.2f} C {tm.parse_ms:.2f} D {tm.blocks_ms:.2f} E {tm.encode_ms:.2f}] tok={tm.tokens} blk={tm.blocks} unmerged={tm.ranges_unmerged} fb={tm.fallback_walks}")
    if not ok:
        M = O.Model(data, level)
        link = np.zeros(n+8, np.uint16); m2 = np.zeros(n+8, np.uint32); mq = np.zeros(n+8, np.uint32)
        tok = np.zeros(n+8, np.uint32); nt = ctypes.c_size_t(0)
        L.szl_engine_debug_fetch(eng, link.ctypes.data, m2.ctypes.data, mq.ctypes.data, n, tok.ctypes.data, n, ctypes.byref(nt))
        bad = np.nonzero(link[:n] != M.link[:n])[0]
        print('   link mismatches', bad.size, bad[:5], link[bad[:5]], M.link[bad[:5]])
        bad = np.nonzero(m2[:n] != M.m2[:n])[0]
        print('   m2 mismatches', bad.size, bad[:5], [hex(x) for x in m2[bad[:5]]], [hex(x) for x in M.m2[bad[:5]]])
        bad = np.nonzero(mq[:n] != M.mq[:n])[0]
        print('   mq mismatches', bad.size, bad[:5], [hex(x) for x in mq[bad[:5]]], [hex(x) for x in M.mq[bad[:5]]])
        rt = tr['tokens']
        print('   ntok', nt.value, rt.size)
        k = min(nt.value, rt.size)
        bad = np.nonzero(tok[:k] != rt[:k])[0]
        print('   token mismatches', bad.size, bad[:5])
        # first differing byte
        m = min(len(comp), len(ref))
        a = np.frombuffer(comp[:m], np.uint8); b = np.frombuffer(ref[:m], np.uint8)
        bad = np.nonzero(a != b)[0]
        rows = np.zeros(8*64, np.uint64); nr = ctypes.c_size_t(0)
        L.szl_engine_debug_blocks(eng, rows.ctypes.data, 64, ctypes.byref(nr))
        print('   gpu blocks', [tuple(int(v) for v in rows[8*i:8*i+8]) for i in range(min(nr.value,4))])
        print('   ref blocks', [(x['type'], x['last'], x['ntokens'], x['bit_start'], x['opt_len'], x['static_len'], x['stored_len']) for x in tr['blocks'][:4]])
        print('   first byte diff', bad[:5], 'of', m, 'blocks', [(x['type'], x['ntokens'], x['bit_start']) for x in tr['blocks'][:6]])
    return ok
allok = True
tests = [('hello', np.frombuffer(b'Hello, world', np.uint8)), ('empty', np.zeros(0,np.uint8)), ('one', np.frombuffer(b'x',np.uint8)),
         ('a32', np.frombuffer(b'a'*32,np.uint8)), ('abc10', np.frombuffer(b'abc'*10,np.uint8)),
         ('dickens', C.generate('dickens',1,0,300000)), ('enwik', C.generate('enwik',2,0,1000000)), ('logs', C.generate('logs',3,0,400000)),
         ('random', C.random_bytes(100000)), ('zeros', C.zeros(150000)), ('acgt', C.four_symbol(200000)), ('p10', C.period10(100000)), ('mixed', C.mixed(700000))]
for name, d in tests:
    allok &= run(name, d, 6)
for lv in (5,7,8,9):
    allok &= run('mixed', C.mixed(700000), lv)
    allok &= run('enwik', C.generate('enwik',2,0,500000), lv)
print('ALL OK' if allok else 'SOME FAILED')
