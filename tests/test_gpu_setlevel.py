"""SetLevel / SetStrategy in mid-stream (run with -m gpu): the streaming Deflater against the oracle's on random switch points.

Reference: C/Deflater.cs:349-395 (SetLevel, SetStrategy) and C/DeflaterEngine.cs:304-361 (DeflaterEngine.SetLevel): within
one compression function (levels 1-4 DeflateFast, 5-9 DeflateSlow) a call only replaces goodLength / max_lazy / niceLength /
max_chain (and the strategy) for the iterations the engine has not run yet — and, once the caller has drained Deflate()
until it returns 0, the engine stands where its loop stopped for want of lookahead: the first iteration start less than
MIN_LOOKAHEAD = 262 bytes before the end of the input it was given.  The device parses a segment only at Flush()/Finish(),
so it reproduces that position from the call history (a caller that changes the level while compressed bytes are still
waiting in the reference's pending buffer gets the position of the drained pattern: DESIGN §7).
Bit-exact, tolerance 0.
"""
import numpy as np
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def _drain(d, o, got, ref, buf):
    while True:
        k = d.Deflate(buf)
        if k <= 0:
            break
        got += buf[:k].tobytes()
    while True:
        b = o.deflate(8192)
        if not b:
            break
        ref += b


def _run(levels, seed, total=600000, strategies=(0,), flush_p=0.25, call_before_deflate_p=0.2, max_calls_per_segment=None,
         cross_kind_at_flush=False, cross_p=0.0, chunk_sizes=(1, 3, 100, 261, 262, 263, 700, 5000, 40000, 70000), stats=None):
    """cross_p: share of the SetLevel calls that may pick ANY of `levels` — another compression function (DeflateStored /
    DeflateFast / DeflateSlow) with bytes pending: the reference flushes a block with the old function where its engine stands."""
    from sharpziplib_amd.deflater import Deflater
    rng = np.random.default_rng(seed)
    data = np.concatenate([C.generate("enwik", seed, 0, total // 2), C.generate("logs", seed + 1, 0, total - total // 2)])
    level = int(rng.choice(levels))
    d, o = Deflater(level, True), O.Deflater(level, True)
    d.CallerDrains()                                      # (these drivers take all Deflate() offers before every change: include/szl.h)
    got, ref = bytearray(), bytearray()
    buf = np.zeros(8192, np.uint8)
    pos, calls = 0, 0
    log = []
    kind = lambda lv: 0 if lv == 0 else (1 if lv < 5 else 2)

    def maybe_switch():
        nonlocal calls
        if max_calls_per_segment is not None and calls >= max_calls_per_segment:
            return True
        r = rng.random()
        if r < 0.45:
            if rng.random() < cross_p:
                lv = int(rng.choice(levels))
            else:
                lv = int(rng.choice([l for l in levels if kind(l) == kind(d_level[0])]))   # the same compression function
            d.SetLevel(lv)
            o.set_level(lv)
            if lv != d_level[0]:
                calls += 1
                if stats is not None and kind(lv) != kind(d_level[0]):
                    k = "%d>%d" % (kind(d_level[0]), kind(lv))
                    stats[k] = stats.get(k, 0) + 1
            d_level[0] = lv
            log.append(("level", pos, lv))
        elif r < 0.6 and len(strategies) > 1:
            st = int(rng.choice(strategies))
            if st != d_strat[0]:
                calls += 1
            d.SetStrategy(st); o.set_strategy(st)
            d_strat[0] = st
            log.append(("strategy", pos, st))
        return True

    def drain_both():
        while True:                                # drain until Deflate() returns 0: only then has the reference's engine run
            k = d.Deflate(buf)                     # as far as its lookahead allows (it pauses after every block while output
            if k <= 0:                             # is pending, C/DeflaterEngine.cs:126-139; IsNeedingInput alone says nothing
                break                              # about that) — the position the device assumes, see DESIGN §7
            got.extend(buf[:k].tobytes())
        while True:
            b = o.deflate(8192)
            if not b:
                break
            ref.extend(b)

    d_level, d_strat = [level], [0]
    while pos < data.size:
        n = int(rng.choice(chunk_sizes))
        c = data[pos:pos + n]
        pos += c.size
        d.SetInput(c); o.set_input(c)
        if rng.random() < call_before_deflate_p:
            if not maybe_switch():                 # the engine has not seen this chunk yet
                return
        drain_both()
        assert d.IsNeedingInput and o.needs_input
        if not maybe_switch():                     # the engine stopped within 261 bytes of the end of this chunk
            return
        if rng.random() < 0.5:
            drain_both()                           # the block a function change flushed (the next Deflate() call hands it out anyway)
            assert bytes(ref).startswith(bytes(got)) or bytes(got).startswith(bytes(ref)), (seed, pos, log[-6:])
        if rng.random() < flush_p:
            d.Flush(); o.flush()
            drain_both()
            calls = 0
            assert bytes(got) == bytes(ref), (seed, pos, log[-6:])
            if cross_kind_at_flush and rng.random() < 0.6:      # another function right after a flush
                lv = int(rng.choice(levels))
                d.SetLevel(lv)
                o.set_level(lv)
                d_level[0] = lv
                log.append(("level@flush", pos, lv))
    d.Finish(); o.finish()
    while not d.IsFinished:
        k = d.Deflate(buf)
        assert k > 0
        got += buf[:k].tobytes()
    while not o.finished:
        ref += o.deflate(8192)
    assert bytes(got) == bytes(ref), (seed, log[-8:])
    assert d.TotalIn == o.total_in == data.size


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_slow_levels_random_switch_points(seed):
    _run([5, 6, 7, 8, 9], seed)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_slow_levels_with_strategy_changes(seed):
    _run([5, 6, 9], seed, strategies=(0, 1, 2))


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_fast_levels_random_switch_points(seed):
    _run([1, 2, 3, 4], seed, strategies=(0, 2))


@pytest.mark.parametrize("seed", [41, 42, 43, 44, 45, 46])
def test_fast_and_slow_levels_alternate_at_flushes(seed):
    """SetLevel across DeflateFast / DeflateSlow right after Flush(): the new function walks the hash chains the old one left —
    DeflateFast does not insert the inside of long matches (C/DeflaterEngine.cs:697-712), DeflateSlow inserts everything."""
    _run([1, 2, 3, 4, 5, 6, 7, 9], seed, flush_p=0.4, cross_kind_at_flush=True)


def test_switch_without_any_flush():
    """one segment from the first byte to Finish(), four parameter changes inside"""
    _run([5, 6, 8, 9], 31, total=900000, flush_p=0.0, max_calls_per_segment=4)


def test_many_switches_inside_one_segment():
    """a dozen parameter changes between two flushes (the reference takes any number, C/DeflaterEngine.cs:304-361)"""
    _run([5, 6, 7, 8, 9], 32, total=500000, flush_p=0.03, call_before_deflate_p=0.5, chunk_sizes=(100, 700, 5000, 20000))


@pytest.mark.parametrize("seed", [51, 52, 53, 54, 55, 56, 57, 58])
def test_fast_and_slow_levels_switch_with_bytes_pending(seed):
    """DeflateFast <-> DeflateSlow anywhere: the old function flushes a block where its engine stands (FlushBlock(.., false),
    C/DeflaterEngine.cs:335-353; a pending lazy match of DeflateSlow is dropped and its first byte tallied), the new one goes on
    from there on the hash chains the old one left."""
    _run([1, 2, 3, 4, 5, 6, 7, 9], seed, cross_p=0.7, total=400000)


@pytest.mark.parametrize("seed", [61, 62, 63, 64, 65, 66, 67, 68, 69, 70])
def test_all_three_functions_switch_with_bytes_pending(seed):
    """level 0 in the mix: FlushStoredBlock of what DeflateStored consumed (:327-333; its UpdateHash() is redone by FillWindow :396),
    stored blocks that start inside a byte after a coded block, DeflateStored taking over a coded function's lookahead."""
    st = {}
    _run([0, 1, 3, 4, 5, 6, 9], seed, cross_p=0.8, total=400000, stats=st)
    print("switches", st)


@pytest.mark.parametrize("seed", [81, 82, 83, 84])
def test_all_three_functions_small_chunks_cross_window_bases(seed):
    """small SetInput chunks, few flushes: chunk ends fall everywhere relative to the window base, the engine slides between them"""
    st = {}
    _run([0, 2, 6], seed, cross_p=0.9, total=300000, flush_p=0.02, chunk_sizes=(1, 50, 261, 262, 700, 3000, 9000), stats=st)
    print("switches", st)

