"""Differential tests of the device Inflater on corrupted and hand-assembled streams (run with -m gpu).

For every stream the device result — status, bytes produced, bytes consumed — is compared with the oracle's
(oracle/szl_inflate_oracle.c, the C restatement of C/Inflater.cs + C/InflaterHuffmanTree.cs + C/InflaterDynHeader.cs):
  * stream decodes on the oracle  -> the device must return the same bytes and the same `in_consumed`;
  * stream fails on the oracle    -> the device must fail with the status that maps to the same exception, and the bytes
    the reference had delivered before throwing (asking one byte per Inflate() call: the longest prefix any caller can
    have seen) must be a prefix of what the device produced.  The device decodes ahead of the caller, so it may hold more
    bytes (everything before the bad token) — never different ones.
"""
import numpy as np
import pytest
import zlib

import corrupt_streams as CS
import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu

# oracle error -> szl_status (include/szl.h); -100 = more output than the cap, -102 = "Unexpected EOF" (CS/InflaterInputStream.cs:494)
ERR = {-1: -16, -2: -17, -3: -18, -4: -19, -5: -20, -6: -21, -7: -22, -8: -23, -9: -24, -10: -26, -13: -27, -100: -6, -102: -25}
CAP = 192 * 1024


@pytest.fixture(scope="module")
def eng():
    from sharpziplib_amd.batch import Engine
    e = Engine()
    yield e
    e.close()


QUIRKS = []   # streams that hold a code set on which the reference's table is not a canonical decoder (exact-table mode on the device)


def _compare(name, stream, status, out, consumed, failures):
    n, delivered, cons = O.inflate_probe(stream, max_out=CAP)
    if O.inflate_probe.quirk_sets:
        # The stream holds an INCOMPLETE code-length set with codes of 10+ bits: the reference's lookup table then differs from
        # every canonical decoder (C/InflaterHuffmanTree.cs:153-163,200-203), and an entry of it can make StreamManipulator drop more
        # bits than it holds (CS/StreamManipulator.cs:86-90).  The device decodes such a block with k_inflate_exact — that very
        # table (csrc/szl_inflate_reftree.h) and that very bit buffer — compared like every other stream; only counted here.
        QUIRKS.append(name)
    if n >= 0:
        if status != 0 or out != delivered or consumed != cons:
            failures.append("%s: oracle ok (%d bytes, consumed %d) but device status %d, %d bytes, consumed %d%s" % (
                name, n, cons, status, len(out), consumed, "" if out == delivered else " [bytes differ]"))
        return
    want = ERR.get(n, None)
    if status != want:
        failures.append("%s: oracle error %d (-> %s) but device status %d (%d bytes)" % (name, n, want, status, len(out)))
    elif out[:len(delivered)] != delivered:
        failures.append("%s: bytes delivered before error %d differ (oracle %d, device %d)" % (name, n, len(delivered), len(out)))


def _run_batch(eng, cases):
    res = eng.inflate([s if len(s) else b"" for _, s in cases], [CAP] * len(cases))
    failures = []
    for (name, s), (r, consumed) in zip(cases, res):
        _compare(name, s, r.status, r.data, consumed, failures)
    return failures


def _run_streaming(cases):
    """The streaming object (C/Inflater.cs member set): SetInput(all), Inflate() in 4 KiB calls."""
    from sharpziplib_amd.deflater import SharpZipBaseException
    from sharpziplib_amd.inflater import Inflater
    from sharpziplib_amd import _lib
    failures = []
    L = _lib.lib()
    for name, s in cases:
        inf = Inflater(True)
        if len(s):
            inf.SetInput(s)
        out = bytearray()
        buf = np.zeros(4096, np.uint8)
        status = 0
        while True:
            room = min(4096, CAP - len(out))
            if room == 0:
                k = L.szl_inflater_inflate(inf._h, buf.ctypes.data, 1)   # any further output = more than the cap
                status = -6 if k > 0 else (k if k < 0 else (0 if inf.IsFinished else -25))
                break
            k = L.szl_inflater_inflate(inf._h, buf.ctypes.data, room)
            if k < 0:
                status = k
                break
            out += buf[:k].tobytes()
            if inf.IsFinished:
                break
            if k == 0:
                status = -25 if inf.IsNeedingInput else -103
                break
        consumed = inf.TotalIn
        _compare(name + " [streaming]", s, status, bytes(out), consumed, failures)
    return failures


def test_crafted_streams_batch(eng):
    fails = _run_batch(eng, CS.crafted())
    assert not fails, "\n".join(fails)


def test_crafted_streams_streaming_object():
    fails = _run_streaming(CS.crafted())
    assert not fails, "\n".join(fails)


def test_distance_before_start_yields_zeros(eng):
    """a match reaching before the first output byte reads the zeros of a fresh OutputWindow
    (CS/OutputWindow.cs:22,63-92), never memory in front of the stream's output region (here: a canary-filled neighbour)."""
    cases = [c for c in CS.crafted() if c[0].startswith("dist_before_start")]
    canary = O.deflate(np.full(70000, 0xFD, np.uint8), 0)        # its output region sits right in front of each probe's
    bufs, caps = [], []
    for _, s in cases:
        bufs += [canary, s]; caps += [70000, CAP]
    res = eng.inflate(bufs, caps)
    for i, (name, s) in enumerate(cases):
        r, consumed = res[2 * i + 1]
        n, delivered, cons = O.inflate_probe(s, max_out=CAP)
        assert n >= 0 and r.status == 0 and r.data == delivered and consumed == cons, name
        assert b"\xfd" not in r.data, name


def _valid_streams():
    rng = np.random.default_rng(20260921)
    v = []
    for kind, seed, n in (("dickens", 11, 6000), ("enwik", 12, 20000), ("logs", 13, 12000)):
        d = C.generate(kind, seed, 0, n)
        v.append(("%s_L6" % kind, O.deflate(d, 6)))
        v.append(("%s_L1" % kind, O.deflate(d, 1)))
        v.append(("%s_L9_flush" % kind, O.deflate(d, 9, flush=True)))
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        v.append(("%s_zlib" % kind, co.compress(d.tobytes()) + co.flush()))
    v.append(("random_stored", O.deflate(C.random_bytes(9000, seed=5), 6)))
    v.append(("level0", O.deflate(C.generate("dickens", 14, 0, 70000), 0)))
    v.append(("tiny_static", O.deflate(np.frombuffer(b"hello hello hello hello", np.uint8), 6)))
    v.append(("zeros", O.deflate(C.zeros(50000), 6)))
    return v, rng


def test_bitflips_and_truncations_batch(eng):
    valid, rng = _valid_streams()
    cases = valid + CS.mutations(valid, rng, n_flip=12, n_trunc=4)
    fails = _run_batch(eng, cases)
    assert not fails, "%d of %d streams differ:\n%s" % (len(fails), len(cases), "\n".join(fails[:40]))


def test_bitflips_and_truncations_streaming_object():
    valid, rng = _valid_streams()
    cases = CS.mutations(valid[:6], rng, n_flip=4, n_trunc=2)
    fails = _run_streaming(cases)
    assert not fails, "%d of %d streams differ:\n%s" % (len(fails), len(cases), "\n".join(fails[:40]))


def test_header_region_flips(eng):
    """Flips concentrated in the first 60 bytes: block type, HLIT/HDIST/HCLEN, the code-length code and its RLE stream —
    the part of a stream where over-subscribed / incomplete sets and repeat errors come from."""
    valid, rng = _valid_streams()
    cases = []
    for name, s in valid:
        b = np.frombuffer(s, np.uint8)
        for k in range(24):
            pos = int(rng.integers(0, min(b.size, 60) * 8))
            m = b.copy(); m[pos >> 3] ^= 1 << (pos & 7)
            cases.append(("%s_hdrflip@%d" % (name, pos), m.tobytes()))
    fails = _run_batch(eng, cases)
    assert not fails, "%d of %d streams differ:\n%s" % (len(fails), len(cases), "\n".join(fails[:40]))


def test_code_sets_the_reference_table_decodes_differently(eng):
    """incomplete sets with codes of 10+ bits followed by random bits, whole and truncated — the device must go
    through the reference's own table quirks (unassigned second-level slots = "symbol 0, 0 bits", long codes of the last
    partial prefix written into the primary table, IndexOutOfRange out of the constructor) and arrive at the same bytes,
    status and `in_consumed`."""
    rng = np.random.default_rng(0xA16A16)
    cases = CS.crafted_long_code_incomplete() + CS.quirk_set_streams(rng, 900)
    QUIRKS.clear()
    fails = _run_batch(eng, cases)
    assert not fails, "%d of %d streams differ:\n%s" % (len(fails), len(cases), "\n".join(fails[:40]))
    assert len(QUIRKS) > 300                 # the class is what this test is about


def test_code_sets_the_reference_table_decodes_differently_streaming_object():
    rng = np.random.default_rng(0xA16A17)
    cases = CS.crafted_long_code_incomplete() + CS.quirk_set_streams(rng, 150)
    fails = _run_streaming(cases)
    assert not fails, "%d of %d streams differ:\n%s" % (len(fails), len(cases), "\n".join(fails[:40]))
