"""Stage B's instruction text on the CPU (no GPU): csrc/szl_match9_asm.h — the whole main loop of k_match9 and its tail program, hand-
written gfx950 assembly — runs in tools/wavesim.py on a tile staged the way the kernel stages it, 16 interleaved wavefronts, and its
match tables must equal oracle/szl_model.c's (FindLongestMatch for every position, C/DeflaterEngine.cs:474-612).  The simulator also
checks what the assembler cannot: a register read with its LDS load still in flight, an LDS access out of range, an unknown
instruction.  (Small tiles: the device suite covers the sizes that matter; this keeps the text honest between GPU runs.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tools", "sim_match9.py")


@pytest.mark.parametrize("args", [
    ["--kib", "96", "--tlen", "3072", "--tile", "3"],                                  # text, level 6, the tail program with the move of B's walks
    ["--kind", "logs", "--level", "9", "--kib", "64", "--tlen", "2048", "--tile", "4"],  # long chains, nice length 258
    ["--kind", "dickens", "--level", "5", "--kib", "96", "--tlen", "4096", "--tile", "2", "--mth", "-1"],   # no move: the two-context tail loop to the end
    ["--kib", "96", "--tlen", "3072", "--tile", "5", "--cut", "100"],                  # the segment ends inside the tile's lookahead: clamps of cap / nice
    ["--kib", "96", "--tlen", "3072", "--tile", "3", "--tailp", "0"],                  # the main loop to the end (the laboratory form)
    ["--kind", "zeros", "--kib", "64", "--tlen", "2048", "--tile", "3"],
    # round 6: the plain one-context loop (the default is the run-ahead loop), slices to the end, the first tile (history before the stream)
    ["--kib", "96", "--tlen", "3072", "--tile", "3", "--tailp", "1", "--ktail1", "1", "--guide", "0"],
    ["--kind", "logs", "--level", "9", "--kib", "64", "--tlen", "4096", "--tile", "0", "--ktail1", "2"],
    ["--kind", "logs", "--level", "6", "--kib", "96", "--tlen", "6144", "--tile", "7", "--ktail1", "1", "--guide", "100000"],
    ["--kind", "dickens", "--level", "8", "--kib", "128", "--tlen", "8192", "--tile", "9"],
    # form 1 of the text (SZL9_V 1: the first filter byte follows the walk's last failed compare)
    ["--adapt", "--kind", "logs", "--level", "9", "--kib", "64", "--tlen", "4096", "--tile", "5"],
    ["--adapt", "--kind", "logs", "--level", "6", "--kib", "64", "--tlen", "4096", "--tile", "0"],
    ["--adapt", "--kib", "96", "--tlen", "3072", "--tile", "5", "--cut", "100"],
    ["--adapt", "--kind", "dickens", "--level", "7", "--kib", "96", "--tlen", "4096", "--tile", "2", "--mth", "-1"],
    ["--adapt", "--kib", "96", "--tlen", "3072", "--tile", "3", "--tailp", "1", "--ktail1", "1"],
    ["--adapt", "--kind", "zeros", "--kib", "64", "--tlen", "2048", "--tile", "3"],
])
def test_instruction_text_reproduces_the_model(args):
    r = subprocess.run([sys.executable, SIM] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "m2 mismatches 0, mq mismatches 0" in r.stdout and "stray stores 0" in r.stdout, r.stdout[-1500:]
