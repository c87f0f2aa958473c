"""Call patterns where a backend that compresses at Flush()/Finish() — or decodes ahead of the caller — could part from the reference
without anybody noticing (run with -m gpu).  Round 4 documented three; each is now either exact or fails loudly, and each
has a test that constructs the pattern against the oracle:
  * Reset() with input the engine has not taken (C/DeflaterEngine.cs:234-253 keeps inputBuf): exact;
  * a block the reference decodes non-canonically, fed in pieces of odd length (CS/StreamManipulator.cs:244-262): exact while the phase
    of the reference's 16-bit loads can be known, SZL_E_UNSUPPORTED -> NotSupportedException otherwise — never other garbage;
  * SetLevel while compressed bytes still wait in the reference's pending buffer (C/DeflaterEngine.cs:126-139): the device equals the
    reference for a caller who drains Deflate() (the only pattern the reference's own stream classes have) and has said so
    (szl_deflater_caller_drains; the stream classes do); any other caller's change is refused (round 6: no environment variable
    needed), since the object cannot see which of the two it is talking to; the test pins both of the reference's byte strings."""
import numpy as np
import pytest

import corrupt_streams as CS
import oracle_ffi as O
from sharpziplib_amd import _lib
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _drain_both(d, o, size=8192):
    got, ref = bytearray(), bytearray()
    buf = np.zeros(size, np.uint8)
    while True:
        k = d.Deflate(buf)
        if k <= 0:
            break
        got += buf[:k].tobytes()
    while True:
        b = o.deflate(size)
        if not b:
            break
        ref += b
    return bytes(got), bytes(ref)


# ---- Reset() with input the engine has not consumed ---------------------------------------------------------------------------------
@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("nowrap", [True, False])
def test_reset_keeps_input_no_deflate_call_has_followed(level, nowrap):
    from sharpziplib_amd.deflater import Deflater, InvalidOperation
    a = C.generate("enwik", 61, 0, 50000)
    b = C.generate("logs", 62, 0, 30011)
    d, o = Deflater(level, nowrap), O.Deflater(level, nowrap)
    d.SetInput(a); o.set_input(a)
    _drain_both(d, o)                                        # drained: the engine has taken `a` (what the reference hands out before a flush —
    #                                                          full blocks — this backend hands out AT the flush: not compared here)
    d.SetInput(b); o.set_input(b)                            # ... and `b` it has not seen when Reset() arrives
    d.Reset(); o.reset()
    assert d.IsNeedingInput == o.needs_input == False       # noqa: E712 — the input is still there (:186-189)
    with pytest.raises(InvalidOperation, match="Old input was not completely processed"):
        d.SetInput(a[:10])                                   # (:163-166)
    assert o.set_input(a[:10]) < 0
    d.Finish(); o.finish()
    got, ref = _drain_both(d, o)
    assert got == ref == O.deflate(b, level, nowrap=nowrap)  # the next stream opens with the bytes that were left
    assert d.TotalIn == o.total_in == b.size and d.IsFinished
    # and once more through Flush + further input, to see that the carried bytes are ordinary input of the new stream
    d.Reset(); o.reset()
    d.SetInput(a); o.set_input(a)
    d.Reset(); o.reset()
    d.Flush(); o.flush()
    g1, r1 = _drain_both(d, o)
    d.SetInput(b); o.set_input(b); d.Finish(); o.finish()
    g2, r2 = _drain_both(d, o)
    assert g1 == r1 and g2 == r2


def test_reset_after_a_drained_deflate_carries_nothing():
    from sharpziplib_amd.deflater import Deflater
    a = C.generate("dickens", 63, 0, 40000)
    d, o = Deflater(6, True), O.Deflater(6, True)
    d.SetInput(a); o.set_input(a)
    _drain_both(d, o)
    d.Reset(); o.reset()
    assert d.IsNeedingInput and o.needs_input
    d.SetInput(a[:5000]); o.set_input(a[:5000]); d.Finish(); o.finish()
    got, ref = _drain_both(d, o)
    assert got == ref == O.deflate(a[:5000], 6)


# ---- the exact decoder and pieces of odd length -------------------------------------------------------------------------------------
def _feed(inf_set_input, inf_inflate, needs_input, finished, pieces, cap=1 << 20):
    out = bytearray()
    it = iter(pieces)
    inf_set_input(next(it))
    while len(out) < cap:
        b = inf_inflate(4096)
        out += b
        if finished():
            break
        if not b:
            if not needs_input():
                break
            nxt = next(it, None)
            if nxt is None:
                break
            inf_set_input(nxt)
    return bytes(out)


def _device(pieces):
    from sharpziplib_amd.inflater import Inflater
    inf = Inflater(True)
    buf = np.zeros(4096, np.uint8)

    def infl(n):
        k = inf.Inflate(buf, 0, n)
        return buf[:k].tobytes()
    try:
        out = _feed(inf.SetInput, infl, lambda: inf.IsNeedingInput, lambda: inf.IsFinished, pieces)
        return "ok", out, (inf.TotalIn if inf.IsFinished else None)    # (TotalIn of a stream that ran dry counts the reference's bit buffer: compared at IsFinished, as everywhere)
    except Exception as e:                                    # noqa: BLE001 — the type is what is compared
        return type(e).__name__, None, None


def _oracle(pieces):
    inf = O.Inflater(True)

    def infl(n):
        k, b = inf.inflate(n)
        if k < 0:
            raise RuntimeError("oracle status %d" % k)
        return b
    try:
        out = _feed(inf.set_input, infl, lambda: inf.needs_input, lambda: inf.finished, pieces)
        return "ok", out, (inf.total_in if inf.finished else None)
    except Exception as e:                                    # noqa: BLE001
        return type(e).__name__, None, None


def test_quirk_blocks_fed_in_pieces_are_exact_or_refused():
    rng = np.random.default_rng(0xD1CE5)
    streams = [s for _, s in CS.quirk_set_streams(rng, 240) if len(s) > 40]
    exact = refused = 0
    for s in streams:
        for cuts in ((len(s) // 2 & ~1,), (17, 17 + 64), (len(s) // 3 | 1,), (9, 30, 51)):
            cuts = [c for c in cuts if 0 < c < len(s)]
            pieces = [s[a:b] for a, b in zip([0] + cuts, cuts + [len(s)])]
            want = _oracle(pieces)
            got = _device(pieces)
            later_odd = any(len(p) & 1 for p in pieces[1:])
            if got[0] == "NotSupportedOnDevice":
                assert later_odd, "refused although every later piece has even length"
                refused += 1
                continue
            if want[0] == "ok":
                assert got == want, "pieces %s: device %s/%s vs reference %s/%s" % ([len(p) for p in pieces], got[0], got[2], want[0], want[2])
            else:
                assert got[0] != "ok", "the reference throws (%s), the device decoded" % want[0]
            exact += 1
    assert exact > 300 and refused > 0, (exact, refused)


# ---- SetLevel while compressed bytes wait in the reference's pending buffer ----------------------------------------------------------
def test_setlevel_with_output_pending_is_exact_for_a_declared_drainer_and_refused_otherwise():
    from sharpziplib_amd.deflater import Deflater, NotSupportedOnDevice
    data = C.generate("enwik", 64, 0, 400000)
    a, b = data[:300000], data[300000:]

    def reference(drain):
        o = O.Deflater(6, True)
        o.set_input(a)
        out = bytearray(o.deflate(512))                       # ONE call with DeflaterOutputStream's buffer size: blocks are left in `pending`
        if drain:
            while True:
                x = o.deflate(512)
                if not x:
                    break
                out += x
        o.set_level(9)                                        # where the engine stands now depends on how much output was taken
        while not o.needs_input:
            x = o.deflate(512)
            if not x:
                break
            out += x
        o.set_input(b); o.finish()
        while not o.finished:
            out += o.deflate(65536)
        return bytes(out)
    drained, undrained = reference(True), reference(False)
    assert drained != undrained                               # the pattern exists: the reference's bytes depend on the caller's buffer
    buf = np.zeros(512, np.uint8)
    big = np.zeros(1 << 20, np.uint8)

    def device(declare):
        d = Deflater(6, True)
        if declare:
            d.CallerDrains()                                  # what the stream classes say for the Deflater they drive (include/szl.h)
        d.SetInput(a)
        assert d.Deflate(buf) == 0                            # this backend compresses at Flush() / Finish(): "nothing yet"
        d.SetLevel(9)
        d.SetInput(b); d.Finish()
        got = bytearray()
        while not d.IsFinished:
            k = d.Deflate(big)
            got += big[:k].tobytes()
        return bytes(got)
    # no environment variable, no declaration: the object cannot know which of the two callers it has, and says so instead of guessing
    with pytest.raises(NotSupportedOnDevice):
        device(False)
    assert device(True) == drained                            # == the reference for the caller who drains (CS/DeflaterOutputStream.cs:242-272)
    d = Deflater(6, True)                                     # ... and nothing is refused while nothing could have been produced yet
    d.SetInput(a[:3000]); d.Deflate(buf); d.SetLevel(9)
    d = Deflater(6, True)                                     # ... or after a Flush(): the engine stands at the end of the input then, whoever calls
    d.SetInput(a); d.Flush()
    while d.Deflate(big):
        pass
    d.SetLevel(9)
    # SZL_STRICT=0: the silent assumption of rounds 3-5
    L = _lib.lib()
    L.szl_debug_set(b"SZL_STRICT", 0)
    try:
        assert device(False) == drained
    finally:
        L.szl_debug_set(b"SZL_STRICT", -2147483648)
