"""gfxsim.same_code — which kernels' MACHINE CODE differs from a given git revision (default: HEAD)?

    python tools/gfxsim/same_code.py [rev]

Every kernel translation unit is compiled to assembly at `rev` (sources out of `git show`, headers of that revision too) and in the working
tree (same flags as csrc/Makefile); kernel bodies are compared instruction by instruction with labels and symbol names normalised.  For
working without a device: a change that is meant to leave the shipped kernels alone (a knob-gated variant, a refactoring of host code) can
show that it does — one more template instantiation in a unit is enough to change the inliner's decisions for the kernels next to it
(round 4: a second k_spec_win instantiation changed seven neighbours).  Test infrastructure only.
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sharpziplib_amd", "csrc")
UNITS = ["szl_kernels_match", "szl_kernels_match2", "szl_kernels_match9", "szl_kernels_fast", "szl_kernels_parse", "szl_kernels_block",
         "szl_kernels_checksum", "szl_kernels_inflate", "szl_kernels_inflate_exact", "szl_kernels_inflate_par", "szl_engine", "szl_api", "szl_api_inflate"]


def bodies(path):
    t = open(path).read()
    out = {}
    for m in re.finditer(r"\n(_Z[A-Za-z0-9_]*):[^\n]*\n; %bb.0:", t):
        i = m.end()
        j = t.find("s_endpgm", i)
        if j < 0:
            continue
        b = re.sub(r"\.LBB\d+_", ".LBB_", t[i:j])
        b = re.sub(r"_Z[A-Za-z0-9_]+", "SYM", b)
        b = re.sub(r";[^\n]*", "", b)
        b = re.sub(r"(?m)^(\s*)\d+:", r"\1N:", b)               # inline assembly's local labels (%= numbers them per function)
        b = re.sub(r"\b\d+([fb])\b", r"N\1", b)
        out[m.group(1)] = b
    return out


def asm_of(srcdir, unit, out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-DSZL_LAB=0",
                           "-I" + os.path.join(srcdir, "include"), os.path.join(srcdir, "sharpziplib_amd", "csrc", unit + ".hip"), "-o", out],
                          stderr=subprocess.DEVNULL)


def main(argv):
    rev = argv[0] if argv else "HEAD"
    tmp = tempfile.mkdtemp(prefix="same_code_")
    old = os.path.join(tmp, "old")
    os.makedirs(os.path.join(old, "sharpziplib_amd", "csrc")); os.makedirs(os.path.join(old, "include"))
    names = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "-r", "--name-only", rev, "sharpziplib_amd/csrc", "include"], text=True).split()
    for n in names:
        if n.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(old, n), "wb") as f:
                f.write(subprocess.check_output(["git", "-C", ROOT, "show", "%s:%s" % (rev, n)]))
    changed = total = 0
    for u in UNITS:
        if not os.path.exists(os.path.join(old, "sharpziplib_amd", "csrc", u + ".hip")):
            print("%-28s not in %s" % (u, rev)); continue
        a, b = os.path.join(tmp, u + ".old.s"), os.path.join(tmp, u + ".new.s")
        asm_of(old, u, a); asm_of(ROOT, u, b)
        A, B = bodies(a), bodies(b)
        norm = lambda k: re.sub(r"Lb[01]E(E+v)", r"\1", k)          # (a defaulted bool template parameter added at the end since: compare by what is left)
        Bn = {}
        for k, v in B.items():
            Bn.setdefault(norm(k), []).append(v)
        diff = []
        for k, v in A.items():
            total += 1
            cands = B.get(k) and [B[k]] or Bn.get(norm(k), [])
            if v not in cands:
                diff.append(k)
        new_only = len(B) - len(A)
        changed += len(diff)
        print("%-28s %3d kernels at %s, %3d now, %d with other machine code%s" % (u, len(A), rev, len(B), len(diff), (": " + ", ".join(subprocess.run(["c++filt"] + diff, capture_output=True, text=True).stdout.split("\n")[:4])[:200]) if diff else ""))
    print("%d of %d kernels of %s have other machine code in the working tree" % (changed, total, rev))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
