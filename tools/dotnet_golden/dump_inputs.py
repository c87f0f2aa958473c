#!/usr/bin/env python3
"""Writes the inputs of the golden cases as files for tools/dotnet_golden/Program.cs (the C# side cannot regenerate the seeded corpora).

    python tools/dotnet_golden/dump_inputs.py <dir> [--headline]

<dir>/cases.tsv: name, file, level, mode — the 29 cases of tests/golden/make_golden.py as raw one-shot streams (the names the JSON
uses: "<case>-L<level>"), the tiny vectors of SURVEY App. C.8, three of them with zlib framing, three through the
DeflaterOutputStream Write / Flush / Finish pattern, and with --headline the streams of tests/golden/headline_golden.json up to 1 GiB
(cfg2_enwik_1g_l6 is what bench.py times; the 2 and 4 GiB log streams with --headline-all)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                                           # noqa: E402

from golden.make_golden import CASES, make_input               # noqa: E402
from golden.make_headline import CASES as HEADLINE             # noqa: E402
from sharpziplib_amd import corpus as C                        # noqa: E402

TINY = {"empty": b"", "x": b"x", "Hello": b"Hello", "Hello_world": b"Hello, world", "testfile": b"testfile contents\n",
        "bytes_0_255": bytes(range(256)), "a32": b"a" * 32, "abc10": b"abc" * 10}


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    rows = []

    def put(fname, data):
        path = os.path.join(out, fname)
        if not os.path.exists(path) or os.path.getsize(path) != len(data):
            with open(path, "wb") as f:
                f.write(data if isinstance(data, (bytes, bytearray)) else memoryview(np.ascontiguousarray(data)))
        return fname
    for name, spec, levels in CASES:
        f = put(name + ".bin", make_input(spec))
        for lv in levels:
            rows.append(("%s-L%d" % (name, lv), f, lv, "raw"))
    for name, data in TINY.items():
        f = put("tiny_" + name + ".bin", data)
        for lv in (0, 6):
            rows.append(("tiny_%s-L%d" % (name, lv), f, lv, "raw"))
    for name in ("dickens_1m", "logs_1m", "zeros_200k"):
        rows.append(("%s-zlib-L6" % name, name + ".bin", 6, "zlib"))
        rows.append(("%s-stream-L6" % name, name + ".bin", 6, "stream"))
    if "--headline" in sys.argv or "--headline-all" in sys.argv:
        for name, (kind, seed, off, n, level, entry) in HEADLINE.items():
            if entry or (n > (1 << 30) and "--headline-all" not in sys.argv):
                continue                                       # (many-entry cases are covered entry by entry in the small cases)
            rows.append((name, put("headline_%s.bin" % name, C.generate(kind, seed, off, n)), level, "raw"))
    with open(os.path.join(out, "cases.tsv"), "w") as f:
        f.write("# name\tfile\tlevel\tmode\n")
        for r in rows:
            f.write("%s\t%s\t%d\t%s\n" % r)
    print("wrote %d cases to %s" % (len(rows), out))


if __name__ == "__main__":
    main()
