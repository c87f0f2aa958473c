"""gfxsim.suite — parity cases of the product's MACHINE CODE on the CPU interpreter, against the oracle.

    python tools/gfxsim/suite.py <suite> [...]        (run from the repo root; `list` names the suites)

Every case goes through the product's C ABI (libszl_amd's own host objects linked against the fake runtime) and the gfx950
assembly of its kernels, interpreted instruction by instruction; the expected bytes come from oracle/ exactly as in the GPU
tests — whose helpers these suites reuse with small sizes.  What the interpreter cannot show: timing, memory-ordering between
wavefronts, hardware behaviour outside the ISA text (DESIGN 4.1's exchange order is MODELLED as ascending here, not proved).
Test infrastructure only; tests/test_sim_product_code.py runs each suite in its own process.
"""
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                   # noqa: E402


def _attach():
    from gfxsim import harness
    rt = harness.use(fast_probe=True, memcheck=bool(os.environ.get("GFXSIM_MEMCHECK")), racecheck=bool(os.environ.get("GFXSIM_RACECHECK")))
    return harness, rt


def _classes():
    from sharpziplib_amd import corpus as C
    return {"dickens": lambda n: C.generate("dickens", 0xD1CE, 0, n), "logs": lambda n: C.generate("logs", 0x106, 0, n),
            "random": lambda n: C.random_bytes(n), "zeros": lambda n: C.zeros(n), "acgt": lambda n: C.four_symbol(n),
            "p10": lambda n: C.period10(n), "mixed": lambda n: C.mixed(n)}


def suite_deflate_levels():
    """the batch entry point, levels 0-9 x data classes, raw and zlib framing, checksums"""
    import oracle_ffi as O
    from sharpziplib_amd.batch import Engine
    e = Engine()
    cl = _classes()
    n = 0
    for name in ("dickens", "logs", "random", "mixed"):
        data = cl[name](2000)
        for lv in range(10):
            r = e.deflate([data], level=lv, crc32=True, adler32=True)[0]
            assert r.status == 0 and r.data == O.deflate(data, lv), (name, lv)
            assert r.crc32 == O.crc32(data) and r.adler32 == O.adler32(data), (name, lv)
            n += 1
    for lv in (1, 6):
        data = cl["dickens"](1500)
        r = e.deflate([data], level=lv, nowrap=False)[0]
        assert r.data == O.deflate(data, lv, nowrap=False) and zlib.decompress(r.data) == data.tobytes()
        n += 1
    e.close()
    return n


def suite_deflate_shapes():
    """several streams in one call, the tiny vectors of the reference's behaviour, sizes around MIN_LOOKAHEAD, strategies"""
    import oracle_ffi as O
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd import corpus as C
    e = Engine()
    n = 0
    for data, hexout in [(b"", "0300"), (b"x", "ab0000"), (b"Hello", "f348cdc9c90700"), (b"Hello, world", "f348cdc9c9d75128cf2fca490100"),
                         (b"a" * 32, "4b240000"), (b"abc" * 10, "4b4c4ac68300"), (b"testfile contents\n", "2b492d2e49cbcc495548cecf2b49cd2b29e60200")]:
        assert e.deflate([data], level=6)[0].data.hex() == hexout, data
        n += 1
    sizes = [0, 1, 2, 3, 4, 261, 262, 263, 700, 1031]
    bufs = [C.generate("dickens", 21 + i, 0, s) if s else np.zeros(0, np.uint8) for i, s in enumerate(sizes)]
    for lv in (1, 6, 9):
        res = e.deflate(bufs, level=lv, crc32=True)
        for b, r in zip(bufs, res):
            assert r.status == 0 and r.data == O.deflate(b, lv) and r.crc32 == O.crc32(b), (lv, b.size)
            n += 1
    data = C.mixed(3000)
    for strategy in (1, 2):
        for lv in (3, 6):
            r = e.deflate([data], level=lv, strategy=strategy)[0]
            assert r.data == O.deflate(data, lv, strategy=strategy), (strategy, lv)
            n += 1
    e.close()
    return n


def suite_deflater_object():
    """the streaming Deflater: SetInput pieces, Flush, SetLevel / SetStrategy in mid-stream (all three compression functions),
    Reset with stale bits, a preset dictionary — tests/test_gpu_setlevel.py's driver with small totals"""
    import oracle_ffi as O
    import test_gpu_setlevel as TS
    import test_gpu_reset_bits as TR
    from sharpziplib_amd import corpus as C
    from sharpziplib_amd.deflater import Deflater
    small = (1, 3, 100, 261, 262, 263, 700, 1500)
    n = 0
    for levels, seed, kw in (([5, 6, 7, 8, 9], 1, {}), ([5, 6, 9], 11, dict(strategies=(0, 1, 2))), ([1, 2, 3, 4], 21, dict(strategies=(0, 2))),
                             ([1, 2, 3, 4, 5, 6, 7, 9], 41, dict(flush_p=0.4, cross_kind_at_flush=True)),
                             ([1, 2, 3, 4, 5, 6, 7, 9], 51, dict(cross_p=0.7)), ([0, 1, 3, 4, 5, 6, 9], 61, dict(cross_p=0.8)),
                             ([0, 2, 6], 81, dict(cross_p=0.9, flush_p=0.05))):
        _knobs(SZL_UP_SLAB_KIB=1 if seed % 20 == 1 else FORGET)     # (the pending bytes travel to the device as they arrive, here in 1 KiB slabs: every other configuration)
        TS._run(levels, seed, total=7000, chunk_sizes=small, **kw)
        n += 1
    _knobs(SZL_UP_SLAB_KIB=FORGET)
    for level, nowrap in ((6, True), (1, False)):
        d, o = Deflater(level, nowrap), O.Deflater(level, nowrap)
        a = C.generate("enwik", 3, 0, 1234)
        d.SetInput(a); o.set_input(a)
        d.Flush(); o.flush()
        got, ref = TR._drain(d, o)
        assert got == ref
        b = C.generate("logs", 53, 0, 2100)
        TR._second_stream(d, o, b, "level %d" % level)
        TR._second_stream(d, o, a[:777], "level %d (after Finish)" % level)
        n += 1
    # preset dictionary (C/Deflater.cs:407-421 -> C/DeflaterEngine.cs:198-229)
    dic = C.generate("dickens", 5, 0, 900)
    data = np.concatenate([dic[300:700], C.generate("dickens", 6, 0, 1500)])
    for level in (1, 6):
        d, o = Deflater(level, False), O.Deflater(level, False)
        d.SetDictionary(dic); o.set_dictionary(dic)
        d.SetInput(data); o.set_input(data)
        d.Finish(); o.finish()
        got, ref = TR._drain(d, o)
        assert got == ref, ("dictionary", level)
        n += 1
    # Deflate() without its copies (szl_deflater_deflate_view: what the device-aware DeflaterOutputStream writes from) — tests/test_gpu_write_path.py
    import test_gpu_write_path as TW
    for level, nowrap in ((0, True), (1, False), (6, True), (6, False)):
        d, o = Deflater(level, nowrap), O.Deflater(level, nowrap)
        got, ref = bytearray(), bytearray()
        for i, part in enumerate((C.generate("enwik", 5, 0, 3001), C.generate("logs", 6, 0, 2222), C.generate("enwik", 7, 0, 1500))):
            d.SetInput(part); o.set_input(part)
            got += TW._views(d); ref += TW._odrain(o)
            if i == 1:
                d.Flush(); o.flush()
                got += TW._views(d); ref += TW._odrain(o)
                assert bytes(got) == bytes(ref) and d.TotalOut == len(got), ("view at a flush", level)
        d.Finish(); o.finish()
        buf = np.zeros(7, np.uint8)
        k = d.Deflate(buf)                                      # a few bytes through the copying call, the rest in place
        got += buf[:k].tobytes() + TW._views(d); ref += TW._odrain(o)
        assert bytes(got) == bytes(ref) and d.IsFinished and d.TotalOut == len(got) and d.DeflateView() is None, ("view", level)
        n += 1
    return n


def suite_inflate():
    """the batch Inflater and the streaming object on streams of several encoders and block mixes; gzip / zlib framing"""
    import oracle_ffi as O
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd import corpus as C
    e = Engine()
    n = 0
    text = C.generate("dickens", 0xD1CE, 0, 6000)
    members = []
    for lv in (0, 1, 6, 9):
        co = zlib.compressobj(lv, zlib.DEFLATED, -15)
        members.append((text, co.compress(text.tobytes()) + co.flush()))
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    members.append((text, co.compress(text.tobytes()) + co.flush()))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)                        # flush mix: sync-flushed pieces, a stored block in between
    mix = co.compress(text[:2000].tobytes()) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(text[2000:4000].tobytes()) + co.flush(zlib.Z_FULL_FLUSH) + \
        co.compress(text[4000:].tobytes()) + co.flush()
    members.append((text, mix))
    for name in ("logs", "random", "zeros", "mixed"):
        d = _classes()[name](3000)
        members.append((d, O.deflate(d, 6)))
    res = e.inflate([m for _, m in members], [d.size for d, _ in members], crc32=True)
    for (d, m), (r, used) in zip(members, res):
        assert r.status == 0 and r.data == d.tobytes() and used == len(m) and r.crc32 == zlib.crc32(d.tobytes())
        n += 1
    # trailing bytes are not consumed; zlib framing with Adler-32
    z = zlib.compress(text.tobytes(), 6) + b"TRAILER"
    (r, used), = e.inflate([z], [text.size], nowrap=False, adler32=True)
    assert r.status == 0 and r.data == text.tobytes() and used == len(z) - 7 and r.adler32 == zlib.adler32(text.tobytes())
    n += 1
    # the streaming object: input in pieces, output in small calls, RemainingInput exact
    inf = Inflater(True)
    comp = members[2][1] + b"xyz"
    out = bytearray()
    buf = bytearray(700)
    pos = 0
    while not inf.IsFinished:
        if inf.IsNeedingInput:
            inf.SetInput(comp[pos:pos + 500]); pos += 500
        k = inf.Inflate(buf)
        out += buf[:k]
    assert bytes(out) == text.tobytes() and inf.TotalOut == text.size
    assert inf.TotalIn == len(comp) - 3, (inf.TotalIn, len(comp))
    n += 1
    e.close()
    return n


def suite_inflate_corrupt():
    """hand-assembled and corrupted streams, the code sets the reference's table decodes differently (k_inflate_exact) included:
    status, bytes and in_consumed against the oracle — tests/test_gpu_inflate_fuzz.py's generators and comparison on a sample"""
    import oracle_ffi as O
    import corrupt_streams as CS
    import test_gpu_inflate_fuzz as TF
    from sharpziplib_amd import corpus as C
    from sharpziplib_amd.batch import Engine
    e = Engine()
    TF.CAP = 8192
    rng = np.random.default_rng(20260922)
    valid = []
    for kind, seed, n in (("dickens", 11, 1500), ("logs", 13, 2500)):
        d = C.generate(kind, seed, 0, n)
        valid += [("%s_L6" % kind, O.deflate(d, 6)), ("%s_L1" % kind, O.deflate(d, 1)), ("%s_L9_flush" % kind, O.deflate(d, 9, flush=True))]
    valid.append(("random_stored", O.deflate(C.random_bytes(900, seed=5), 6)))
    valid.append(("tiny_static", O.deflate(np.frombuffer(b"hello hello hello hello", np.uint8), 6)))
    cases = CS.crafted() + valid + CS.mutations(valid, rng, n_flip=6, n_trunc=2)
    for name, s in valid:                                     # flips in the header region (code-length sets, repeat errors)
        b = np.frombuffer(s, np.uint8)
        for k in range(6):
            pos = int(rng.integers(0, min(b.size, 60) * 8))
            m = b.copy(); m[pos >> 3] ^= 1 << (pos & 7)
            cases.append(("%s_hdrflip@%d" % (name, pos), m.tobytes()))
    TF.QUIRKS.clear()
    quirk = CS.crafted_long_code_incomplete() + CS.quirk_set_streams(np.random.default_rng(0xA16A16), 60)
    fails = TF._run_batch(e, cases + quirk)
    assert not fails, "%d of %d differ:\n%s" % (len(fails), len(cases) + len(quirk), "\n".join(fails[:10]))
    assert len(TF.QUIRKS) > 20, len(TF.QUIRKS)                # the class k_inflate_exact exists for
    stream_cases = cases[::5] + quirk[::4]
    fails = TF._run_streaming(stream_cases)
    assert not fails, "\n".join(fails[:10])
    e.close()
    return len(cases) + len(quirk) + len(stream_cases)


def _knobs(**kw):
    from sharpziplib_amd import _lib
    for k, v in kw.items():
        _lib.lib().szl_debug_set(k.encode(), int(v))


FORGET = -2147483648


def suite_forms():
    """the forms the defaults do not take at small sizes: the window pipeline of long streams (64 KiB windows), stage B's other
    form kept in the product library — stage A's bucketed form k_links2, the fallback of the ticket form —, ranges that never merge (zeros / periodic data)"""
    import oracle_ffi as O
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd import corpus as C
    n = 0
    text = C.generate("enwik", 0xE9, 0, 100000)
    try:
        _knobs(SZL_WINDOW_KIB=64, SZL_WINDOW_FROM_KIB=0)
        e = Engine()
        for data, lv in ((text, 6), (C.zeros(90000), 9)):
            r = e.deflate([data], level=lv, crc32=True)[0]
            assert r.status == 0 and r.data == O.deflate(data, lv) and r.crc32 == O.crc32(data), ("window", lv)
            n += 1
        e.close()
    finally:
        _knobs(SZL_WINDOW_KIB=256 * 1024, SZL_WINDOW_FROM_KIB=2048 * 1024)
    small = C.generate("dickens", 3, 0, 24000)
    for knobs, levels in ((dict(SZL_LINKS=2), (6, 9)),):   # (k_match4 and the on-demand walk moved to the laboratory library in round 5: suite lab_forms)
        try:
            _knobs(**knobs)
            e = Engine()
            for lv in levels:
                r = e.deflate([small], level=lv)[0]
                assert r.status == 0 and r.data == O.deflate(small, lv), (knobs, lv)
                n += 1
            e.close()
        finally:
            _knobs(**{k: FORGET for k in knobs})
    e = Engine()
    for data, lv in ((C.zeros(40000), 6), (C.period10(30000), 9), (C.four_symbol(20000), 6)):
        r = e.deflate([data], level=lv)[0]
        assert r.status == 0 and r.data == O.deflate(data, lv), ("never merging", lv)
        n += 1
    e.close()
    return n


def suite_inflate_parallel():
    """one member on many wavefronts (finder, chunk jobs, window chain, convert) with 16 KiB chunks, against the one-wavefront
    decoder and the input; a damaged member steps aside with the reference's status"""
    import oracle_ffi as O
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd import corpus as C
    n = 0
    e = Engine()
    try:
        data = C.generate("enwik", 0xE9, 0, 360000)         # (a member needs 8 chunks of 16 KiB and 4 block starts found to go parallel)
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 4)     # memLevel 4: a block every 1024 tokens
        members = [("zlib6_small_blocks", co.compress(data.tobytes()) + co.flush())]
        for name, m in members:
            assert len(m) > 131072
            _knobs(SZL_INF_CHUNK_KIB=16, SZL_INF_PAR_MIN_KIB=64)
            (r, used), = e.inflate([m], [data.size], crc32=True)
            jobs = int(_lib.lib().szl_engine_debug_par_jobs(e._h))
            assert jobs >= 4, jobs
            assert r.status == 0 and r.data == data.tobytes() and used == len(m) and r.crc32 == zlib.crc32(data.tobytes()), name
            n += 1
        # data that does not compress: runs of stored blocks (the finder names their headers too; a job copies a run without leaving its
        # loop), a member of such stretches and text, and a member whose payload is deflate data — the INNER streams' block headers are
        # false candidate starts the jobs move on past (tests/test_gpu_inflate_par.py)
        rnd = C.random_bytes(200000, seed=3)
        ci = zlib.compressobj(6, zlib.DEFLATED, -15, 4)
        inner = np.frombuffer(ci.compress(data.tobytes()) + ci.flush(), np.uint8)       # a block header every ~1 KiB
        mixed = np.concatenate([rnd[:70000], data[:80000], rnd[70000:140000], data[80000:120000]])
        for name, plain, lv in (("stored_only_l6", rnd, 6), ("mixed", mixed, 6), ("payload_is_deflate_data", inner, 6)):
            m = O.deflate(plain, lv)
            _knobs(SZL_INF_CHUNK_KIB=16, SZL_INF_PAR_MIN_KIB=64)
            (r, used), = e.inflate([m], [plain.size], crc32=True)
            jobs = int(_lib.lib().szl_engine_debug_par_jobs(e._h))
            assert r.status == 0 and r.data == plain.tobytes() and used == len(m) and r.crc32 == zlib.crc32(plain.tobytes()), name
            assert jobs >= 4 or len(m) < 131072, (name, jobs, len(m))
            n += 1
        bad = bytearray(members[0][1]); bad[90000] ^= 0x10
        (rp, up), = e.inflate([bytes(bad)], [data.size])
        _knobs(SZL_INF_PAR_MIN_KIB=1 << 22)
        (rs, us), = e.inflate([bytes(bad)], [data.size])
        assert (rp.status, up, rp.data) == (rs.status, us, rs.data)
        no, delivered, cons = O.inflate_probe(bytes(bad), max_out=data.size)
        assert (no >= 0) == (rs.status == 0)
        n += 1
    finally:
        _knobs(SZL_INF_CHUNK_KIB=FORGET, SZL_INF_PAR_MIN_KIB=FORGET)
        e.close()
    return n


def suite_inflate_dense():
    """(laboratory library, GFXSIM_LAB=1) the symbol pass compiled for 3 wavefronts per SIMD (SZL_INF_DENSE=1: k_inflate<true, 2, true>, another register
    allocation of the same source; measured and not adopted, profiles/r05/dense_ab.log) on one member, against the input; a reference-made member whose blocks are longer
    than the chunks (jobs span several chunks: staging regions by span)"""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd import corpus as C
    e = Engine()
    try:
        data = C.generate("enwik", 0xE9, 0, 360000)
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 4)
        m = co.compress(data.tobytes()) + co.flush()
        _knobs(SZL_INF_CHUNK_KIB=16, SZL_INF_PAR_MIN_KIB=64, SZL_INF_SLOTS_PER_CU=10, SZL_INF_DENSE=1)
        (r, used), = e.inflate([m], [data.size], crc32=True)
        assert int(_lib.lib().szl_engine_debug_par_jobs(e._h)) >= 4
        assert r.status == 0 and r.data == data.tobytes() and used == len(m) and r.crc32 == zlib.crc32(data.tobytes())
        _knobs(SZL_INF_SLOTS_PER_CU=FORGET, SZL_INF_DENSE=FORGET)
        import oracle_ffi as O
        n = 1
        # a reference-made member whose blocks are longer than the 16 KiB chunks (jobs span several chunks; staging regions sized by span)
        small = C.generate("enwik", 0xE9, 0, 360000)
        ms = O.deflate(small, 6)
        (r, used), = e.inflate([ms], [small.size], crc32=True)
        assert int(_lib.lib().szl_engine_debug_par_jobs(e._h)) >= 3
        assert r.status == 0 and r.data == small.tobytes() and used == len(ms)
        n += 1
    finally:
        _knobs(SZL_INF_CHUNK_KIB=FORGET, SZL_INF_PAR_MIN_KIB=FORGET, SZL_INF_SLOTS_PER_CU=FORGET, SZL_INF_DENSE=FORGET)
        e.close()
    return n


def suite_multi_device():
    """two (fake) devices behind the C ABI: streams split into groups (szl_deflate_batch_multi_host / szl_inflate_batch_multi_host, one
    host thread and engine per device), ONE stream cut into position-range units with warm-up and hand-over check
    (stream_multi_run), a bad ordinal, szl_multi_release — the host logic of SURVEY 8e with distinct ordinals"""
    import oracle_ffi as O
    from gfxsim import harness
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import deflate_multi, inflate_multi
    from sharpziplib_amd import corpus as C
    harness._state["fake"].fakehip_set_device_count(2)
    assert _lib.lib().szl_device_count() == 2
    n = 0
    rng = np.random.default_rng(7)
    bufs = [C.generate(("dickens", "logs", "enwik")[i % 3], 500 + i, 0, int(rng.integers(0, 2500))) for i in range(7)]
    for devices in ([0, 1], [1, 0, 1]):
        res = deflate_multi(bufs, devices, level=6, crc32=True)
        for b, r in zip(bufs, res):
            assert r.status == 0 and r.data == O.deflate(b, 6) and r.crc32 == O.crc32(b), devices
        n += 1
    comp = [r.data for r in res]
    back = inflate_multi(comp, [b.size for b in bufs], [1, 0], crc32=True)
    for b, c, (r, used) in zip(bufs, comp, back):
        assert r.status == 0 and r.data == b.tobytes() and used == len(c) and r.crc32 == O.crc32(b)
    n += 1
    res = deflate_multi([bufs[1], np.zeros(0, np.uint8)], [0, 1, 0, 1], level=9)          # fewer streams than device slots, an empty stream
    assert [r.data for r in res] == [O.deflate(bufs[1], 9), O.deflate(np.zeros(0, np.uint8), 9)]
    n += 1
    try:
        deflate_multi([b"abc"], [0, 2])
        raise AssertionError("ordinal 2 of 2 devices was accepted")
    except _lib.SzlError:
        n += 1
    try:
        _knobs(SZL_PART_MIN_KIB=64, SZL_WINDOW_KIB=64, SZL_PART_WARM_KIB=64)
        data = C.generate("enwik", 0x5EED, 0, 140000)
        (r,) = deflate_multi([data], [0, 1], level=6, crc32=True)
        assert r.status == 0 and r.crc32 == O.crc32(data) and r.data == O.deflate(data, 6)
        n += 1
    finally:
        _knobs(SZL_PART_MIN_KIB=FORGET, SZL_WINDOW_KIB=FORGET, SZL_PART_WARM_KIB=FORGET)
    assert _lib.lib().szl_multi_release() == 0
    res = deflate_multi(bufs[:3], [1, 0], level=6)
    assert [r.data for r in res] == [O.deflate(b, 6) for b in bufs[:3]]
    n += 1
    return n


def suite_exchange_order():
    """fault injection (GFXSIM_XCHG=desc | flaky:<per mille>:<after n exchanges>): the interpreter serves ds_wrxchg in another lane order than
    DESIGN 4.1 assumes of the chip — always (the probe must refuse k_links3) or now and then after the probe has passed (the check every
    exchange carries must notice, the call is run again with k_links2).  Whatever happens, the bytes are the oracle's."""
    import oracle_ffi as O
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd import corpus as C
    mode = os.environ.get("GFXSIM_XCHG", "")
    assert mode, "run with GFXSIM_XCHG=desc or flaky:<per mille>:<after>"
    from gfxsim import harness
    rt = harness._state["rt"]
    e = Engine()
    rng = np.random.default_rng(int(os.environ.get("GFXSIM_XCHG_SEED", "165")))
    gens = [lambda n: C.zeros(n), lambda n: C.period10(n), lambda n: C.four_symbol(n), lambda n: np.tile(np.frombuffer(b"abcabcabd", np.uint8), n // 9 + 1)[:n],
            lambda n: C.generate("logs", 5, 0, n), lambda n: C.mixed(n, seed=3)]
    n = 0
    for k in range(12):
        data = gens[k % len(gens)](int(rng.integers(300, 4000)))
        lv = (1, 6, 9, 3, 5)[k % 5]
        r = e.deflate([data], level=lv)[0]
        assert r.status == 0 and r.data == O.deflate(data, lv), (mode, k, lv)
        n += 1
    l3 = sum(1 for a, b, c in rt.launches if "k_links3" in a)
    l2 = sum(1 for a, b, c in rt.launches if "k_links2" in a)
    trips = e.timing()["links_guard_trips"]
    print("exchange order %s: k_links3 launches %d, k_links2 launches %d, guard trips %d" % (mode, l3, l2, trips))
    if mode == "desc":
        assert l3 == 0 and l2 == 12
    e.close()
    return n


def suite_inflate_stream_bulk():
    """the streaming Inflater given a long input (InflaterInputStream with a large buffer): SetInput of 256 KiB or more goes to the
    chunk-parallel decoder from the carried window (inflater_bulk) — zlib framing, input in two pieces, reads of 1 byte to 200 KB,
    RemainingInput / Adler exact; then the device-aware read path: GZipInputStream over the read-ahead InflaterInputBuffer (pinned buffer,
    SetInput without a host copy, CRC-32 kept on the device) on two members and trailing garbage"""
    import gzip
    import io
    from sharpziplib_amd import _lib
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd import corpus as C
    from sharpziplib_amd import gzipstream as G
    data = C.generate("enwik", 0xE9, 0, 400000)
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 4)          # memLevel 4: a block every 1024 tokens
    z = co.compress(data.tobytes()) + co.flush() + b"tail"
    cut = 136 * 1024
    assert len(z) > cut + 4096
    n = 0
    try:
        _knobs(SZL_INF_CHUNK_KIB=16, SZL_INF_PAR_MIN_KIB=64, SZL_INF_STREAM_BULK_KIB=128)
        inf = Inflater(False)
        inf.SetInput(z[:cut])
        out = bytearray()
        sizes = [1, 4096, 70000, 7, 120000]
        k = 0
        while not inf.IsFinished:
            if inf.IsNeedingInput:
                inf.SetInput(z[cut:])
            buf = bytearray(sizes[k % len(sizes)]); k += 1
            out += buf[:inf.Inflate(buf)]
            if k == 3:
                assert inf.Adler == zlib.adler32(bytes(out))     # (in mid-stream: the checksum of what has been handed out, not of what was decoded)
        assert bytes(out) == data.tobytes(), len(out)
        assert inf.RemainingInput == 4 and inf.TotalIn == len(z) - 4 and inf.Adler == zlib.adler32(data.tobytes())
        assert _lib.lib().szl_inflater_debug_bulk_calls(inf._h) >= 1
        n += 1
        # the device-aware classes: 136 KiB of read-ahead (the library's 16 MiB at this suite's scale), a pinned RawData, device CRC
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 4)
        m1 = co.compress(data.tobytes()) + co.flush()
        small = b"second member " * 40
        src = m1 + gzip.compress(small, 9) + b"\0\0garbage"
        class NoSeek(io.RawIOBase):                       # a base stream that cannot tell how long it is: the buffer keeps its full read-ahead,
            def __init__(self, b):                        # Fill() fills it to the brim and promises the Inflater more (szl_inflater_expect_more):
                self.b, self.p = b, 0                     # the piece ends on its last block boundary, the remainder waits for the next Fill()

            def readable(self):
                return True

            def seekable(self):
                return False

            def readinto(self, mv):
                k = min(len(mv), 100000, len(self.b) - self.p)
                mv[:k] = self.b[self.p:self.p + k]
                self.p += k
                return k
        st = G.GZipInputStream(NoSeek(src), 4096, readAhead=136 << 10)
        assert st.inputBuffer._pin is not None and st.inputBuffer.RawData.size == 136 << 10
        assert st.read_all(chunk=150001) == data.tobytes() + small
        assert _lib.lib().szl_inflater_debug_bulk_calls(st.inf._h) >= 1
        st.Dispose()
        cut = src[:len(m1) - 30000]                       # truncated inside the first member: every byte it holds, then "Unexpected EOF"
        st = G.GZipInputStream(NoSeek(cut), 4096, readAhead=136 << 10)
        got, buf = bytearray(), np.zeros(5000, np.uint8)
        try:
            while True:
                k = st.Read(buf, 0, buf.size)
                assert k > 0
                got += buf[:k].tobytes()
        except Exception as e:
            assert "Unexpected EOF" in str(e), e
        want = zlib.decompressobj(31).decompress(cut)      # (the Read() that meets the end throws and takes the bytes it had gathered with it — as the reference's does)
        assert len(want) > 300000 and len(want) - 5000 < len(got) <= len(want) and bytes(got) == want[:len(got)], (len(got), len(want))
        n += 1
        for dev_crc in (True, False):                                             # (small members: the CRC-32 on the device and the reference's way)
            two = gzip.compress(small, 6) + gzip.compress(small[::-1], 9)
            assert G.GZipInputStream(io.BytesIO(two), deviceCrc=dev_crc).read_all(chunk=100) == small + small[::-1]
            bad = bytearray(two); bad[len(gzip.compress(small, 6)) - 8] ^= 1      # the first member's CRC-32 trailer
            try:
                G.GZipInputStream(io.BytesIO(bytes(bad)), deviceCrc=dev_crc).read_all(chunk=100)
                raise AssertionError("a wrong CRC-32 went unnoticed")
            except G.GZipException as e:
                assert "crc sum mismatch" in str(e)
            n += 1
    finally:
        _knobs(SZL_INF_CHUNK_KIB=FORGET, SZL_INF_PAR_MIN_KIB=FORGET, SZL_INF_STREAM_BULK_KIB=FORGET, SZL_INF_BULK_OUT_MIB=FORGET)
    return n


def suite_framing():
    """gzip members on the device (SZL_F_GZIP: header, CRC-32 / ISIZE trailer) through GZipOutputStream / GZipInputStream and the batch
    member writer / reader; a zip archive around one batched call (zipbatch) read back by Python's zipfile and by the device; zlib framing
    with a preset dictionary on both sides (FDICT, NEED_DICT); a sync flush"""
    import datetime
    import gzip
    import io
    import struct
    import zipfile
    import oracle_ffi as O
    from sharpziplib_amd import corpus as C
    from sharpziplib_amd import gzipstream as G
    from sharpziplib_amd import zipbatch as Z
    from sharpziplib_amd.deflater import Deflater
    from sharpziplib_amd.inflater import Inflater
    n = 0
    data = C.generate("enwik", 3, 0, 5000)
    for level in (1, 6):
        bio = io.BytesIO()
        g = G.GZipOutputStream(bio)
        g.IsStreamOwner = False
        g.SetLevel(level)
        g.FileName = "some/dir/h\xe9llo.txt"
        g.ModifiedTime = 1234567890
        for a in range(0, data.size, 1701):
            g.Write(data[a:a + 1701])
        g.Finish()
        got = bio.getvalue()
        hdr = bytes([0x1F, 0x8B, 8, 8]) + (1234567890).to_bytes(4, "little") + bytes([0, 255]) + "h\xe9llo.txt".encode("latin-1") + b"\0"
        assert got == hdr + O.deflate(data, level) + zlib.crc32(data.tobytes()).to_bytes(4, "little") + data.size.to_bytes(4, "little"), level
        assert gzip.decompress(got) == data.tobytes()
        n += 1
    two = got + gzip.compress(b"second member " * 40, 9) + b"\0\0garbage"
    st = G.GZipInputStream(io.BytesIO(two))
    assert st.read_all(chunk=900) == data.tobytes() + b"second member " * 40
    n += 1
    datas = [C.generate("logs", 40 + i, 0, 700 + 300 * i) for i in range(4)] + [np.zeros(0, np.uint8)]
    members = G.write_members(datas, level=6, names=["a.log", None, "c", None, "empty"], mtimes=[1, 2, 3, 4, 5])
    for m, d in zip(members, datas):
        assert gzip.decompress(m) == d.tobytes()
    back = G.read_members(members)
    assert [g for g, _ in back] == [d.tobytes() for d in datas] and [nm for _, nm in back] == ["a.log", None, "c", None, "empty"]
    n += 1
    # zip archive: local headers / descriptors / central directory around ONE device call
    rng = np.random.default_rng(9)
    ents = [("dir/f%02d.txt" % i, C.generate(("dickens", "logs", "enwik")[i % 3], 70 + i, 0, int(rng.integers(0, 3000)))) for i in range(9)]
    z = Z.write_zip(ents, level=6, when=datetime.datetime(2025, 6, 1, 8, 0, 0))
    zf = zipfile.ZipFile(io.BytesIO(z))
    assert zf.testzip() is None
    for info, (name, d) in zip(zf.infolist(), ents):
        sig, _, _, _, _, _, csize, _, nlen, xlen = struct.unpack_from("<IHHHIIIIHH", z, info.header_offset)
        p0 = info.header_offset + 30 + nlen + xlen
        assert info.filename == name and info.CRC == zlib.crc32(d.tobytes()) and z[p0:p0 + csize] == O.deflate(d, 6), name
    assert Z.read_zip(z) == [(nm, d.tobytes()) for nm, d in ents]
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w", zipfile.ZIP_DEFLATED, compresslevel=9) as zw:
        for name, d in ents[:5]:
            zw.writestr(name, d.tobytes())
    assert Z.read_zip(bio.getvalue()) == [(nm, d.tobytes()) for nm, d in ents[:5]]
    n += 2
    # preset dictionary through zlib framing: the Deflater sets FDICT, the Inflater asks for the dictionary (C/Inflater.cs:211-249,:606-620)
    dic = C.generate("dickens", 5, 0, 1200)
    msg = np.concatenate([dic[200:900], C.generate("dickens", 6, 0, 1800)])
    d = Deflater(6, False)
    d.SetDictionary(dic)
    d.SetInput(msg); d.Finish()
    buf = np.zeros(8192, np.uint8)
    k = d.Deflate(buf)
    comp = buf[:k].tobytes()
    o = O.Deflater(6, False); o.set_dictionary(dic); o.set_input(msg); o.finish()
    assert comp == o.deflate(8192)
    zd = zlib.decompressobj(zdict=dic.tobytes())
    assert zd.decompress(comp) == msg.tobytes()
    inf = Inflater(False)
    inf.SetInput(comp)
    out = bytearray(4000)
    assert inf.Inflate(out) == 0 and inf.IsNeedingDictionary
    inf.SetDictionary(dic)
    got = bytearray()
    while not inf.IsFinished:
        k = inf.Inflate(out)
        assert k > 0
        got += out[:k]
    assert bytes(got) == msg.tobytes() and inf.Adler == zlib.adler32(msg.tobytes())
    n += 1
    # a sync flush in mid-stream: the bytes so far decode on their own
    d = Deflater(9, True); o = O.Deflater(9, True)
    d.SetInput(data[:2500]); o.set_input(data[:2500]); d.Flush(); o.flush()
    k = d.Deflate(buf)
    first = buf[:k].tobytes()
    assert first == o.deflate(8192) and zlib.decompressobj(-15).decompress(first) == data[:2500].tobytes()
    d.SetInput(data[2500:]); o.set_input(data[2500:]); d.Finish(); o.finish()
    k = d.Deflate(buf)
    assert buf[:k].tobytes() == o.deflate(8192)
    n += 1
    return n


def suite_lab_forms():
    """the LABORATORY library (GFXSIM_LAB=1: libszl_amd_lab.so's objects) on the interpreter: stage B's dropped forms — chain compression
    (SZL_MATCH_KERNEL=3), the ring (4), the bucket-order search (5), k_match4 (SZL_B9=0), the on-demand walk k_match_lazy — and k_spec_win's four-ranges-per-store write-back (SZL_SPEC_WB=1),
    each against the oracle (tests/test_gpu_stage_b_forms.py's subject, small)"""
    import oracle_ffi as O
    from sharpziplib_amd.batch import Engine
    from sharpziplib_amd import corpus as C
    assert os.environ.get("GFXSIM_LAB"), "run with GFXSIM_LAB=1"
    n = 0
    data = C.generate("dickens", 3, 0, 12000)
    for knobs, levels in ((dict(SZL_MATCH_KERNEL=3), (6,)), (dict(SZL_MATCH_KERNEL=4, SZL_STRIPE_MIN=1, SZL_STRIPE_KIB=64), (9,)),
                          (dict(SZL_MATCH_KERNEL=5), (6,)), (dict(SZL_SPEC_WB=1), (5, 9)), (dict(SZL_B9=0), (6, 9)), (dict(SZL_MATCH_MODE=1), (9,)),
                          (dict(SZL_EMIT_COPY_LAB=0), (6,))):
        try:
            _knobs(**{k: v for k, v in knobs.items() if k != "SZL_EMIT_COPY_LAB"})
            e = Engine()
            if "SZL_MATCH_MODE" in knobs:
                e.debug_match_mode(1)                      # the on-demand walk (k_match_lazy), forced
            for lv in levels:
                r = e.deflate([data], level=lv)[0]
                assert r.status == 0 and r.data == O.deflate(data, lv), (knobs, lv)
                n += 1
            e.close()
        finally:
            _knobs(**{k: FORGET for k in knobs if k != "SZL_EMIT_COPY_LAB"})
    e = Engine()
    _knobs(SZL_SPEC_WB=1)
    try:
        for d2, lv in ((C.generate("logs", 5, 0, 30000), 6), (C.mixed(25000, seed=5), 9), (C.zeros(20000), 6)):
            r = e.deflate([d2], level=lv)[0]
            assert r.data == O.deflate(d2, lv), ("write-back", lv)
            n += 1
    finally:
        _knobs(SZL_SPEC_WB=FORGET)
    e.close()
    return n


SUITES = {"deflate_levels": suite_deflate_levels, "deflate_shapes": suite_deflate_shapes, "deflater_object": suite_deflater_object,
          "inflate": suite_inflate, "inflate_corrupt": suite_inflate_corrupt,
          "forms": suite_forms, "inflate_parallel": suite_inflate_parallel, "inflate_dense": suite_inflate_dense, "multi_device": suite_multi_device, "exchange_order": suite_exchange_order, "inflate_stream_bulk": suite_inflate_stream_bulk, "framing": suite_framing, "lab_forms": suite_lab_forms}


def main(argv):
    if not argv or argv[0] == "list":
        print("\n".join(SUITES))
        return 0
    harness, rt = _attach()
    rc = 0
    for name in argv:
        t = time.time()
        try:
            n = SUITES[name]()
            instr = sum(sum(v.values()) for v in rt.stats.values())
            print("ok %s: %d cases, %.0f s, %d kernel launches, %.1f M wave instructions interpreted" % (name, n, time.time() - t, len(rt.launches), instr / 1e6), flush=True)
        except Exception:
            import traceback
            traceback.print_exc()
            for msg in harness.errors()[-3:]:
                print("  [gfxsim] " + msg)
            print("FAILED %s" % name, flush=True)
            rc = 1
    if rt.mem.shadow is not None or rt.racecheck:
        rep = rt.memcheck_report()
        print("memcheck: %d distinct findings" % len(rep))
        for kind, kern, line, cnt, addr in rep:
            print("  %-36s %6d x  %s (assembly line %d), first at 0x%x" % (kind, cnt, kern[:70], line, addr))
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
