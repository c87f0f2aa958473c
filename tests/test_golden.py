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
