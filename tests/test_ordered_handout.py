"""CPU emulation of the two-pass hand-out of k_match4o (csrc/szl_match4_body.inc, SZL_M4_ORD: positions whose hash-chain head is
near first, the others last, so that a tile ends on short walks).  The device code examines a slice 64 positions at a time,
compacts the positions of the pass's class with one ds_permute (lane r receives the index of the r-th of them), gives them to
the free lanes in rank order and moves the slice pointer just past the last position taken.  This restates that arithmetic
lane by lane and checks the two properties the kernel relies on: every tile position is handed out exactly once, and a wave
never hands out a second-class position before its first pass is over."""
import os
import random

import numpy as np
import pytest


def _emulate(tlen, slice_, th, nwaves, seed):
    rng = random.Random(seed)
    link = [rng.choice([0xFFFF, rng.randrange(1, 32768), rng.randrange(1, 600)]) for _ in range(tlen)]
    counter = [0, 0]                       # one position counter per pass, in LDS on the device
    handed = [0] * tlen
    waves = [dict(wnext=0, wend=0, pas=0, exhausted=False, dead=False, busy=[0] * 128) for _ in range(nwaves)]
    live = nwaves
    while live:
        w = rng.choice([x for x in waves if not x["dead"]])
        for i in range(128):               # some walks end
            if w["busy"][i] and rng.random() < 0.4:
                w["busy"][i] = 0
        for ctx in (0, 1):                 # fetch_ord(A), fetch_ord(B)
            if w["exhausted"]:
                break
            idle = [l for l in range(64) if not w["busy"][ctx * 64 + l]]
            ni = len(idle)
            while ni > 0:
                if w["wnext"] >= w["wend"]:
                    base = counter[w["pas"]]
                    counter[w["pas"]] += slice_
                    w["wnext"], w["wend"] = min(base, tlen), min(base + slice_, tlen)
                    if w["wnext"] >= w["wend"]:
                        if w["pas"] == 0:
                            w["pas"], w["wnext"], w["wend"] = 1, 0, 0
                            continue
                        w["exhausted"] = True
                        break
                cnt = min(64, w["wend"] - w["wnext"])
                want = [lane < cnt and ((link[w["wnext"] + lane] < th) == (w["pas"] == 0)) for lane in range(64)]
                nm = sum(want)
                if nm == 0:
                    w["wnext"] += cnt
                    continue
                recv = [None] * 64         # ds_permute: lane `slot` receives this lane's index; the slots are a permutation
                for lane in range(64):
                    cj = sum(want[:lane])
                    slot = cj if want[lane] else nm + (lane - cj)
                    assert recv[slot] is None
                    recv[slot] = lane
                ntake = min(nm, ni)
                for r, lane in enumerate(idle[:ntake]):      # ds_bpermute: the lane of rank r reads recv of lane r
                    p = w["wnext"] + recv[r]
                    handed[p] += 1
                    assert (link[p] < th) == (w["pas"] == 0)
                    w["busy"][ctx * 64 + lane] = 1
                idle, ni = idle[ntake:], ni - ntake
                w["wnext"] = w["wnext"] + recv[ntake - 1] + 1 if nm > ntake else w["wnext"] + cnt
        if w["exhausted"] and not any(w["busy"]):
            w["dead"] = True
            live -= 1
    return handed


def test_every_position_is_handed_out_once_in_class_order():
    for seed in range(120):
        rng = random.Random(1000 + seed)
        tlen = rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 1000, 2688, 5376, 21504])
        handed = _emulate(tlen, rng.choice([64, 128, 256, 512]), rng.choice([1, 300, 4096, 40000, 70000]), rng.choice([1, 2, 16]), seed)
        assert all(h == 1 for h in handed), (seed, tlen)


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("SZL_TEST_UNVALIDATED") != "1",
                    reason="k_match4o has not run on a device yet (written after round 2's GPU budget was spent): run it by hand, "
                           "under a timeout, with SZL_TEST_UNVALIDATED=1 before it joins the suite")
def test_ordered_kernel_is_bit_exact():
    import oracle_ffi as O
    from sharpziplib_amd import _lib, corpus as C
    from sharpziplib_amd.batch import Engine
    L = _lib.lib()
    data = [C.generate("enwik", 0xE9, 0, 1500000), C.generate("logs", 0x106, 0, 900000), C.four_symbol(300000), C.zeros(150000),
            C.random_bytes(120000, seed=3), C.mixed(1200000, seed=5), np.zeros(0, np.uint8), C.random_bytes(5, seed=1)]
    L.szl_debug_set(b"SZL_ORDERED", 1)
    eng = Engine()
    try:
        eng.debug_match_mode(0)
        for th in (4096, 1, 70000):
            L.szl_debug_set(b"SZL_ORDER_TH", th)
            for level in (6, 9):
                for d, r in zip(data, eng.deflate(data, level=level)):
                    assert r.status == 0 and r.data == O.deflate(d, level), (th, level, d.size)
    finally:
        eng.close()
        L.szl_debug_set(b"SZL_ORDERED", -2147483648)
        L.szl_debug_set(b"SZL_ORDER_TH", -2147483648)
