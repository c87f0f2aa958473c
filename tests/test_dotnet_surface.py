"""The C# shim replaces four files of the reference.  Everything else in the reference's assembly must
compile against the replacements, so every public / protected / internal member the originals declare needs a counterpart of the
same kind of name and the same number of parameters, and every `Type.Member` the other sources name statically must exist.
There is no .NET toolchain in this image; this is the mechanical part of "it compiles" that can be checked without one.

The surface of the originals is a committed fixture (tests/golden/dotnet_surface.json: names and arities, written by
tests/golden/make_dotnet_surface.py); where /root/reference exists the fixture is also checked to be current."""
import glob
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_dotnet_surface as S  # noqa: E402

ROOT = os.path.dirname(HERE)
FIXTURE = os.path.join(HERE, "golden", "dotnet_surface.json")


def ours():
    types = {}
    for p in sorted(glob.glob(os.path.join(ROOT, "sharpziplib_amd", "dotnet", "*.cs"))):
        for ty, mem in S.members(open(p, encoding="utf-8").read()).items():
            types.setdefault(ty, set()).update(mem)
    return types


def test_extractor_on_a_small_class():
    src = """
    namespace N { // comment with public int Fake;
      public class A : B {
        public const int X = 1, Y = 2;
        public enum E { P = A.X, Q }
        public A() : this(1, "a,b") { int local = 0; }
        public A(int a, string s) { }
        internal long T => 5;
        protected virtual int P { get { return 1; } set { } }
        public bool Auto { get; set; } = true;
        protected byte[] f;
        protected Codec c = Zip.Get();
        private int hidden;
        public unsafe int M(byte[] b, int o, int n) { if (o < 0) throw new E("public int Bogus(int x)"); return 0; }
        public override async Task W(byte[] b, int o, int c, CancellationToken t) { await x(); }
        int alsoHidden() { return 0; }
      }
    }"""
    m = S.members(src)
    assert m["A"] == {("field", "X", -1), ("field", "Y", -1), ("type", "E", -1), ("ctor", "A", 0), ("ctor", "A", 2),
                      ("property", "T", -1), ("property", "P", -1), ("property", "Auto", -1), ("field", "f", -1), ("field", "c", -1),
                      ("method", "M", 3), ("method", "W", 4)}
    assert m["E"] == {("enumvalue", "P", -1), ("enumvalue", "Q", -1)}


def test_every_member_of_the_replaced_files_has_a_counterpart():
    ref = json.load(open(FIXTURE))
    mine = ours()
    missing = []
    for ty, mem in ref["types"].items():
        if ty not in mine:
            missing.append("type %s" % ty)
            continue
        have = mine[ty]
        names = {(k, n) for k, n, _ in have}
        for kind, name, arity in mem:
            # a constant may be declared `const` here and `static readonly` there and the like: kind must agree for callables,
            # data members (field / property) may stand in for each other only where the reference's is a property
            if (kind, name, arity) in have:
                continue
            if kind == "property" and ("field", name) in names:
                continue
            missing.append("%s %s.%s%s" % (kind, ty, name, "" if arity < 0 else "/%d" % arity))
    assert not missing, "the replacement files lack: " + ", ".join(missing)


def test_every_static_use_elsewhere_in_the_reference_resolves():
    ref = json.load(open(FIXTURE))
    mine = ours()
    bad = []
    for ty, uses in ref["static_uses"].items():
        names = {n for _, n, _ in mine.get(ty, ())}
        for mem, where in uses.items():
            if mem not in names:
                bad.append("%s.%s (used in %s)" % (ty, mem, where))
    assert not bad, "named by the reference's other sources, absent here: " + ", ".join(bad)
    assert "CompressionLevel" in ref["static_uses"]["Deflater"]          # the round-5 compile break stays covered


@pytest.mark.skipif(not os.path.isdir(S.REF_SRC), reason="the reference's sources are not on this box")
def test_fixture_is_current():
    assert json.loads(json.dumps(S.surface_of_reference())) == json.load(open(FIXTURE))
