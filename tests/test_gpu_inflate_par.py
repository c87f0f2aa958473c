"""One deflate member decoded by many wavefronts (run with -m gpu): szl_inflate_batch_* on members of SZL_INF_PAR_MIN_KIB or more.

The parallel decode (csrc/szl_kernels_inflate_par.hip + inflate_member_parallel in csrc/szl_api_inflate.hip) must be
indistinguishable from the one-wavefront decoder, which is the path checked against the oracle (test_gpu_inflate*.py):
same bytes, same in_consumed, same checksums — and for anything unusual (corrupt, truncated, output too small) it must step
aside, so the status and partial output are the sequential decoder's.  Reference behaviour: C/Inflater.cs:283-552.
The knobs shrink the chunks so that a few MiB already give hundreds of chunk boundaries of every kind.
"""
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    e = Engine()
    yield e
    for k, v in (("SZL_INF_CHUNK_KIB", 128), ("SZL_INF_PAR_MIN_KIB", 2048)):
        _lib.lib().szl_debug_set(k.encode(), v)
    e.close()


def _knobs(chunk_kib, min_kib):
    from sharpziplib_amd import _lib
    _lib.lib().szl_debug_set(b"SZL_INF_CHUNK_KIB", chunk_kib)
    _lib.lib().szl_debug_set(b"SZL_INF_PAR_MIN_KIB", min_kib)


def _par_jobs(eng):
    from sharpziplib_amd import _lib
    return int(_lib.lib().szl_engine_debug_par_jobs(eng._h))


def _both(eng, stream, cap, nowrap=True, chunk_kib=16):
    """(parallel result, sequential result, chunk jobs used)"""
    _knobs(chunk_kib, 64)
    rp = eng.inflate([stream], [cap], nowrap=nowrap, crc32=True)[0]
    jobs = _par_jobs(eng)
    _knobs(chunk_kib, 1 << 22)                     # nothing is long enough: the one-wavefront decoder
    rs = eng.inflate([stream], [cap], nowrap=nowrap, crc32=True)[0]
    assert _par_jobs(eng) == 0
    return rp, rs, jobs


def _same(rp, rs):
    (a, ca), (b, cb) = rp, rs
    assert a.status == b.status and ca == cb and a.crc32 == b.crc32 and a.adler32 == b.adler32
    assert a.data == b.data


@pytest.mark.parametrize("kind,level", [("enwik", 6), ("dickens", 9), ("logs", 6), ("logs", 3), ("enwik", 1)])
def test_member_bit_exact(eng, kind, level):
    data = C.generate(kind, 31, 0, 6 << 20)
    stream = O.deflate(data, level)                 # the reference's own blocks (one per 16 383 tokens at most)
    rp, rs, jobs = _both(eng, stream, data.size)
    assert jobs >= 8, jobs
    _same(rp, rs)
    assert rp[0].status == 0 and rp[0].data == data.tobytes() and rp[1] == len(stream)
    assert rp[0].crc32 == zlib.crc32(data.tobytes())


def test_zeros_and_long_runs(eng):
    """distance-1 runs across chunk boundaries: the overlapping copy must carry the 'byte before the chunk' symbol along"""
    parts = []
    for i in range(24):
        parts.append(np.full(1 << 20, i & 1 and 0x41, np.uint8))
        parts.append(C.generate("enwik", 300 + i, 0, 150000))
        parts.append(C.period10(40000) if hasattr(C, "period10") else np.zeros(40000, np.uint8))
    z = np.concatenate(parts)
    stream = O.deflate(z, 6)
    rp, rs, jobs = _both(eng, stream, z.size)
    assert jobs >= 8
    _same(rp, rs)
    assert rp[0].status == 0 and rp[0].data == z.tobytes()


def test_stored_and_flush_blocks(eng):
    """stored blocks (incompressible stretches), empty stored blocks of sync/full flushes, static blocks — as zlib writes them"""
    rng = np.random.default_rng(5)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts, raw = [], []
    for i in range(40):
        if i % 5 == 3:
            d = C.random_bytes(int(rng.integers(20000, 200000)), seed=100 + i).tobytes()
        else:
            d = C.generate(("enwik", "logs", "dickens")[i % 3], 200 + i, 0, int(rng.integers(50000, 400000))).tobytes()
        raw.append(d)
        parts.append(co.compress(d))
        parts.append(co.flush(zlib.Z_SYNC_FLUSH if i % 2 else zlib.Z_FULL_FLUSH))
    parts.append(co.flush())
    stream, data = b"".join(parts), b"".join(raw)
    rp, rs, jobs = _both(eng, stream, len(data))
    assert jobs >= 8
    _same(rp, rs)
    assert rp[0].status == 0 and rp[0].data == data and rp[1] == len(stream)


def test_runs_of_stored_blocks_are_followed_without_leaving_the_copy(eng):
    """data that does not compress comes as runs of stored blocks — 16 KiB each from a level 5-9 encoder, 64 KiB from level 0, empty ones at
    every sync flush: the decoder follows a run inside its copy loop (header, LEN / NLEN, the next header ...), in the chunk jobs and in the
    one-wavefront form alike; what ends a run — a Huffman block, the final block, a damaged LEN / NLEN pair, the end of the input or of the
    output room — is the careful path's, with the oracle's statuses"""
    rnd = C.random_bytes(5 << 20, seed=9)
    text = C.generate("enwik", 41, 0, 3 << 20)
    mix = np.concatenate([rnd[:1 << 20], text[:700000], rnd[1 << 20:3 << 20], text[700000:2 << 20], rnd[3 << 20:], text[2 << 20:]])
    for level in (0, 6, 9):
        stream = O.deflate(mix, level)
        rp, rs, jobs = _both(eng, stream, mix.size)
        _same(rp, rs)
        assert rp[0].status == 0 and rp[0].data == mix.tobytes() and rp[1] == len(stream), level
        # the output room ends inside a run / the input does: both forms stop where the oracle's Inflater stops
        for cap in (mix.size - 1, (1 << 20) + 12345, 40000):
            rp, rs, _ = _both(eng, stream, cap)
            _same(rp, rs)
            assert rp[0].data == mix.tobytes()[:cap]
        for cut in (len(stream) - 3, len(stream) // 2, 70000):
            rp, rs, _ = _both(eng, stream[:cut], mix.size)
            _same(rp, rs)
            assert rp[0].status != 0 and mix.tobytes().startswith(rp[0].data)
    # zlib's own: level 0 with empty stored blocks in between (sync flushes), then a damaged NLEN in the middle of a run
    co = zlib.compressobj(0, zlib.DEFLATED, -15)
    parts = []
    for i in range(60):
        parts.append(co.compress(mix[i * 100000:(i + 1) * 100000].tobytes()))
        parts.append(co.flush(zlib.Z_SYNC_FLUSH))
    parts.append(co.flush())
    stream = b"".join(parts)
    rp, rs, _ = _both(eng, stream, 6000000)
    _same(rp, rs)
    assert rp[0].status == 0 and rp[0].data == mix.tobytes()[:6000000]
    bad = bytearray(stream)
    pos = len(parts[0]) + len(parts[1]) + len(parts[2]) + len(parts[3])          # the third piece's first stored header
    assert bad[pos] in (0, 1)
    bad[pos + 3] ^= 0x40                                                          # NLEN no longer the complement
    rp, rs, _ = _both(eng, bytes(bad), 6000000)
    _same(rp, rs)
    st, delivered, consumed = O.inflate_probe(np.frombuffer(bytes(bad), np.uint8), max_out=6000000)
    assert st < 0 and rp[0].status != 0 and rp[0].data == delivered == mix.tobytes()[:200000]


def test_member_whose_payload_is_deflate_data(eng):
    """a .tar.gz of zips / PNGs / docx: the member carries deflate streams as its payload — the outer encoder stores them (or finds a little
    to gain), and what the block finder finds in the payload are the INNER streams' block headers, complete and consistent, every few dozen
    KiB.  A job that stands at a boundary past such a candidate goes on to the next one; the member stays on the parallel path"""
    inner = b"".join(O.deflate(C.generate("enwik", 400 + i, 0, 1 << 20), 6) for i in range(16))       # ~6 MiB of deflate data
    inner = np.frombuffer(inner, np.uint8)
    for maker in ("oracle", "zlib"):
        if maker == "oracle":
            stream = O.deflate(inner, 6)
        else:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            stream = co.compress(inner.tobytes()) + co.flush()
        rp, rs, jobs = _both(eng, stream, inner.size)
        _same(rp, rs)
        assert rp[0].status == 0 and rp[0].data == inner.tobytes() and rp[1] == len(stream)
        assert jobs >= 8, (maker, jobs)                      # (not abandoned for the one-wavefront decoder)


def test_zlib_framing_and_adler(eng):
    data = C.generate("enwik", 77, 0, 5 << 20).tobytes()
    stream = zlib.compress(data, 6)
    rp, rs, jobs = _both(eng, stream, len(data), nowrap=False)
    assert jobs >= 8
    _same(rp, rs)
    assert rp[0].status == 0 and rp[0].data == data and rp[0].adler32 == zlib.adler32(data) and rp[1] == len(stream)
    bad = bytearray(stream); bad[-1] ^= 1           # wrong Adler-32 in the trailer (C/Inflater.cs:411-414)
    rp, rs, jobs = _both(eng, bytes(bad), len(data), nowrap=False)
    _same(rp, rs)
    assert rp[0].status == -26 or rp[0].status < 0


def test_static_only_member_steps_aside(eng):
    """no dynamic block header anywhere -> no chunk starts -> the ordinary decoder"""
    data = C.generate("logs", 9, 0, 3 << 20).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    stream = co.compress(data) + co.flush()
    rp, rs, jobs = _both(eng, stream, len(data))
    assert jobs == 0
    _same(rp, rs)
    assert rp[0].data == data


def test_errors_are_the_sequential_decoders(eng):
    data = C.generate("enwik", 41, 0, 4 << 20)
    stream = np.frombuffer(O.deflate(data, 6), np.uint8)
    rng = np.random.default_rng(8)
    cases = [("truncated", stream[:stream.size * 2 // 3].tobytes(), data.size), ("cap_too_small", stream.tobytes(), data.size - 100000)]
    for k in range(10):
        pos = int(rng.integers(stream.size // 8, stream.size)) * 8 + int(rng.integers(0, 8))
        m = stream.copy(); m[pos >> 3] ^= 1 << (pos & 7)
        cases.append(("flip@%d" % pos, m.tobytes(), data.size + 65536))
    for name, s, cap in cases:
        rp, rs, jobs = _both(eng, s, cap)
        try:
            _same(rp, rs)
        except AssertionError:
            raise AssertionError(name)


def test_mixed_batch(eng):
    """a long member between short streams in one call"""
    big = C.generate("enwik", 51, 0, 5 << 20)
    smalls = [C.generate("dickens", 60 + i, 0, 30000 + 1000 * i) for i in range(6)]
    streams = [O.deflate(smalls[0], 6), O.deflate(big, 6)] + [O.deflate(s, 6) for s in smalls[1:]]
    want = [smalls[0], big] + smalls[1:]
    _knobs(16, 64)
    res = eng.inflate(streams, [w.size for w in want], crc32=True)
    assert _par_jobs(eng) >= 8
    for (r, consumed), s, w in zip(res, streams, want):
        assert r.status == 0 and r.data == w.tobytes() and consumed == len(s) and r.crc32 == zlib.crc32(w.tobytes())


@pytest.mark.parametrize("single", [0, 1])
def test_single_pass_and_count_first_forms_agree(eng, single):
    """The decoder normally skips the count pass and sizes every job's staging region from the caller's capacity; a member whose
    expansion varies wildly (random bytes next to long runs) overruns a region and is redone count-first.  Both forms, same bytes."""
    from sharpziplib_amd import _lib
    parts = []
    for i in range(6):
        parts.append(C.random_bytes(700000, seed=50 + i))
        parts.append(np.full(6 << 20, 65 + i, np.uint8))
        parts.append(C.generate("logs", 60 + i, 0, 900000))
    data = np.concatenate(parts)
    stream = zlib.compress(data.tobytes(), 6)[2:-4]
    _lib.lib().szl_debug_set(b"SZL_INF_SINGLE_PASS", single)
    try:
        rp, rs, jobs = _both(eng, stream, data.size)
    finally:
        _lib.lib().szl_debug_set(b"SZL_INF_SINGLE_PASS", -2147483648)
    assert jobs >= 8
    _same(rp, rs)
    assert rp[0].status == 0 and rp[0].data == data.tobytes()


def test_several_long_members_in_one_call(eng):
    """the passes of all long members of a call share their launches; a broken one among them falls back alone"""
    rng = np.random.default_rng(4)
    datas = [C.generate(("enwik", "logs", "dickens")[i % 3], 700 + i, 0, int(rng.integers(2 << 20, 5 << 20))) for i in range(9)]
    streams = [zlib.compress(d.tobytes(), 6) if i % 2 else zlib.compress(d.tobytes(), 9) for i, d in enumerate(datas)]
    bad = bytearray(streams[4]); bad[len(bad) // 2] ^= 0x10
    streams[4] = bytes(bad)
    smalls = [zlib.compress(C.generate("enwik", 900 + i, 0, 20000).tobytes(), 6) for i in range(5)]
    all_streams = streams[:3] + smalls[:2] + streams[3:] + smalls[2:]
    caps = [len(zlib.decompress(s)) if s is not streams[4] else datas[4].size for s in all_streams]
    _knobs(16, 64)
    par = eng.inflate(all_streams, caps, nowrap=False, crc32=True)
    jobs = _par_jobs(eng)
    _knobs(16, 1 << 22)
    seq = eng.inflate(all_streams, caps, nowrap=False, crc32=True)
    assert jobs >= 8 * 8
    for a, b in zip(par, seq):
        _same(a, b)
    for (r, consumed), s in zip(par, all_streams):
        if s is streams[4]:
            assert r.status != 0
        else:
            assert r.status == 0 and r.data == zlib.decompress(s) and consumed == len(s)


def test_default_knobs_256mib_member(eng):
    """library defaults on a member of the size they are meant for; the device's own level-6 stream"""
    from sharpziplib_amd import _lib
    _knobs(128, 2048)
    data = C.generate("enwik", 0xE9, 0, 256 << 20)
    comp = eng.deflate([data], level=6, crc32=True)[0]
    assert comp.status == 0
    (r, consumed), = eng.inflate([comp.data], [data.size], crc32=True)
    assert _par_jobs(eng) >= 256
    assert r.status == 0 and consumed == len(comp.data) and r.crc32 == comp.crc32
    assert r.data == data.tobytes()
    ms = eng.timing()["inflate_ms"]
    print("parallel inflate of a 256 MiB member: %.2f ms = %.0f MiB/s (%d chunk jobs)" % (ms, 256e3 / ms, _par_jobs(eng)))


def test_a_call_of_up_to_1_75_rounds_of_streams_is_decoded_in_chunks(eng):
    """round 6: one wavefront per member is at its worst in ONE round of the 8 x CUs slots (2048 x 4 MiB members: 271 ms, in chunks 225), so
    calls of up to 1.75 rounds' worth of streams (3584 on 256 CUs; 1024 through the round's second third) take the chunked form; more
    streams than that go one wavefront per member as before (csrc/szl_api_inflate.hip, szl_inflate_batch_device)"""
    data = C.generate("enwik", 0x51, 0, 1200 * (384 << 10))
    parts = [data[i * (384 << 10):(i + 1) * (384 << 10)] for i in range(1200)]
    comps = [r.data for r in eng.deflate(parts, level=6)]
    _knobs(16, 64)
    out = eng.inflate(comps, [p.size for p in parts], crc32=True)
    jobs = _par_jobs(eng)
    assert jobs >= 2 * 1200          # (a job per chunk in which a block starts: blocks are ~45 KiB of compressed bytes, chunks 16)
    for (r, consumed), p, c in zip(out, parts, comps):
        assert r.status == 0 and consumed == len(c) and r.data == p.tobytes() and r.crc32 == zlib.crc32(p.tobytes())
    many = [comps[i % 1200] for i in range(4000)]
    out = eng.inflate(many, [384 << 10] * 4000)
    assert _par_jobs(eng) == 0
    assert all(r.status == 0 for r, _ in out) and out[3999][0].data == parts[3999 % 1200].tobytes()
