#!/usr/bin/env python3
"""Regenerates tests/golden/headline_golden.json: sha256 of the oracle's raw-deflate output for BASELINE.json's
configs at (or near) their full sizes, so that the `-m gpu` headline parity tests (tests/test_gpu_headline.py) and
bench.py can pin the device output of the very workload they time.

Like deflate_golden.json these are outputs of oracle/ (the reference is managed C# and cannot run here); they freeze
the bits of the headline streams.  Takes a few minutes of single-thread CPU.  Run: python tests/golden/make_headline.py
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_ffi as O  # noqa: E402
from sharpziplib_amd import corpus as C  # noqa: E402

# name -> (kind, seed, offset, bytes, level, entry_bytes or 0)
CASES = {
    "cfg2_enwik_1g_l6": ("enwik", 0xE9, 0, 1 << 30, 6, 0),          # configs[1]: exactly what bench.py times on rank 0
    "cfg2_prefix_384m_l6": ("enwik", 0xE9, 0, 384 << 20, 6, 0),     # bench.py's cpu_baseline sample, also deflated on the device
    "cfg5_logs_512m_l9": ("logs", 0x106, 0, 512 << 20, 9, 0),       # configs[4] at 1/8 size: level 9, max chains, repetitive logs
    "cfg3_4096x64k_l6": ("dickens", 0x21B0, 0, 4096 * 65536, 6, 65536),  # configs[2] at 4096 entries: many small independent streams
    "off4g_enwik_8m_l6": ("enwik", 0xE9, 7 << 20, 8 << 20, 6, 0),   # the stream placed above 2^32 in the arenas (64-bit offsets)
    "cfg5_logs_2g_l9": ("logs", 0x106, 0, 2 << 30, 9, 0),           # configs[4] at half size: 2 GiB takes the library's default window pipeline
    "cfg1_dickens_64m_l6": ("dickens", 0xD1CE, 0, 64 << 20, 6, 0),  # configs[0]: raw Deflater level 6 on 64 MiB of prose
    "cfg5_logs_4g_l9": ("logs", 0x106, 0, 4 << 30, 9, 0),           # configs[4] at FULL size (bench.py times it through the window pipeline)
    "cfg3_100000x64k_enwik_l6": ("enwik", 0x21B0, 0, 100000 * 65536, 6, 65536),   # configs[2] at FULL size, the very entries bench.py compresses: every one of them hashed
    "cfg5_64x64m_logs_l9": ("logs", 0x106, 0, 4 << 30, 9, 64 << 20),  # configs[4] as bench.py --workload cfg5 runs it over N ranks: the 4 GiB as 64 streams ("shard = stream") that change owner in the rebalance
}


def main():
    out = {"_comment": "sha256 of oracle raw-deflate outputs for the headline configs; see make_headline.py", "cases": {}}
    path = os.path.join(HERE, "headline_golden.json")
    if os.path.exists(path) and "--all" not in sys.argv:      # keep what is there, add what is new
        out["cases"] = json.load(open(path))["cases"]
    for name, (kind, seed, off, n, level, entry) in CASES.items():
        if name in out["cases"]:
            continue
        t = time.time()
        data = C.generate(kind, seed, off, n)
        if entry:
            from multiprocessing.pool import ThreadPool
            h = hashlib.sha256()
            total = 0
            crcs = hashlib.sha256()

            def one(i):                                          # (the oracle runs outside the interpreter lock: one entry per host thread)
                d = data[i * entry:(i + 1) * entry]
                return O.deflate(d, level), int(O.crc32(d))
            with ThreadPool(len(os.sched_getaffinity(0))) as pool:
                for comp, crc in pool.imap(one, range(n // entry), chunksize=max(1, min(64, n // entry // (4 * len(os.sched_getaffinity(0)))))):
                    h.update(comp)
                    crcs.update(crc.to_bytes(4, "little"))
                    total += len(comp)
            rec = {"out_len": total, "out_sha256": h.hexdigest(), "crc_sha256": crcs.hexdigest()}
        else:
            comp = O.deflate(data, level)
            rec = {"out_len": len(comp), "out_sha256": hashlib.sha256(comp).hexdigest(), "crc32": int(O.crc32(data))}
        rec.update({"kind": kind, "seed": seed, "offset": off, "n": n, "level": level, "entry": entry})
        out["cases"][name] = rec
        print("%s: %d -> %d bytes in %.1fs" % (name, n, rec["out_len"], time.time() - t), flush=True)
    with open(os.path.join(HERE, "headline_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
