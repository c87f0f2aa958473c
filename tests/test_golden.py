"""The committed golden fixtures (tests/golden) against the oracle (CPU) — the GPU twin is in test_gpu_deflate.py."""
import hashlib
import json
import os

import pytest

import oracle_ffi as O
from golden.make_golden import make_input

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deflate_golden.json")))["cases"]


@pytest.mark.parametrize("case", GOLD, ids=lambda c: "%s-L%d" % (c["name"], c["level"]))
def test_oracle_matches_golden(case):
    data = make_input(tuple(case["spec"]))
    assert hashlib.sha256(data.tobytes()).hexdigest() == case["in_sha256"], "corpus generator drifted"
    comp = O.deflate(data, case["level"])
    assert len(comp) == case["out_len"] and hashlib.sha256(comp).hexdigest() == case["out_sha256"]
    assert O.crc32(data) == case["crc32"] and O.adler32(data) == case["adler32"]


# ---- the REAL reference's outputs, when somebody has run tools/dotnet_golden/make_reference_golden.sh on a box with a .NET SDK ----------
# (tests/golden/reference_golden.json; absent in this image — no dotnet — so the encoder's parity stays "unpinned")
REF_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden.json")


def oracle_output_of_case(data, level, mode):
    """what tools/dotnet_golden/Program.cs does with the reference's classes, done with the oracle's"""
    import numpy as np
    if mode in ("raw", "zlib"):
        return O.deflate(data, level, nowrap=(mode == "raw"))
    assert mode == "stream"
    # DeflaterOutputStream(sink, new Deflater(level, true), 512): Write in 4096-byte pieces, one Flush() before the piece that holds the middle
    d = O.Deflater(level, True)
    a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
    out = bytearray()

    def drain(flushing):                                   # DeflateSyncOrAsync (CS/DeflaterOutputStream.cs:242-272)
        while flushing or not d.needs_input:
            b = d.deflate(512)
            if not b:
                break
            out.extend(b)
    half = a.size // 2
    for o in range(0, a.size, 4096):
        if o <= half < o + 4096:
            d.flush(); drain(True)
        d.set_input(a[o:o + 4096]); drain(False)
    d.finish()
    while not d.finished:
        b = d.deflate(512)
        if not b:
            break
        out.extend(b)
    return bytes(out)


def reference_cases():
    """(name, data, level, mode) exactly as tools/dotnet_golden/dump_inputs.py lists them"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dotnet_golden"))
    import dump_inputs as D
    from golden.make_golden import CASES
    for name, spec, levels in CASES:
        data = make_input(spec)
        for lv in levels:
            yield "%s-L%d" % (name, lv), data, lv, "raw"
        if name in ("dickens_1m", "logs_1m", "zeros_200k"):
            yield name + "-zlib-L6", data, 6, "zlib"
            yield name + "-stream-L6", data, 6, "stream"
    import numpy as np
    for name, b in D.TINY.items():
        for lv in (0, 6):
            yield "tiny_%s-L%d" % (name, lv), np.frombuffer(b, np.uint8), lv, "raw"


def compare_with_reference(ref):
    seen = 0
    for name, data, level, mode in reference_cases():
        if name not in ref:
            continue
        comp = oracle_output_of_case(data, level, mode)
        assert len(comp) == ref[name]["out_len"] and hashlib.sha256(comp).hexdigest() == ref[name]["out_sha256"], \
            "the oracle's bytes differ from the reference's for %s" % name
        seen += 1
    return seen


@pytest.mark.skipif(not os.path.exists(REF_PATH), reason="tests/golden/reference_golden.json is absent: no .NET SDK in this image; "
                    "tools/dotnet_golden/make_reference_golden.sh writes it on a box that has one (encoder parity stays 'unpinned' until then)")
def test_oracle_equals_the_reference_itself():
    ref = json.load(open(REF_PATH))["cases"]
    assert compare_with_reference(ref) >= 40
    head = json.load(open(os.path.join(os.path.dirname(REF_PATH), "headline_golden.json")))["cases"]
    for name, c in ref.items():                           # headline streams: the frozen oracle hashes against the reference's
        if name in head:
            assert (c["out_len"], c["out_sha256"]) == (head[name]["out_len"], head[name]["out_sha256"]), name


def test_the_reference_comparison_itself_works(tmp_path):
    """the comparator and the case list, exercised with a stand-in file made by the oracle (so that the day the real file appears the only
    thing that can fail is the bytes): every case of dump_inputs.py is found, a changed hash is caught, tiny vectors equal SURVEY App. C.8"""
    ref = {name: {"out_len": len(c), "out_sha256": hashlib.sha256(c).hexdigest()}
           for name, data, level, mode in reference_cases() for c in [oracle_output_of_case(data, level, mode)]}
    assert len(ref) == 51
    assert compare_with_reference(ref) == 51
    assert bytes.fromhex("4b240000") == oracle_output_of_case(b"a" * 32, 6, "raw")          # discriminates against zlib
    # the stream pattern on "Hello": Write + Flush + Finish (T/Base/InflaterDeflaterTests.cs:64-69; SURVEY C.8's f2 48 cd c9 c9 07 08 20 c0 00
    # is Write, Flush, Finish — here the flush comes before the piece that holds the middle, i.e. before the only Write)
    bad = dict(ref); k = next(iter(bad)); bad[k] = dict(bad[k], out_sha256="0" * 64)
    with pytest.raises(AssertionError):
        compare_with_reference(bad)
