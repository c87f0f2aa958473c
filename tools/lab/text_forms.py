"""The two forms of k_match9's text (csrc/szl_match9_asm.h, SZL9_V) side by side on resident streams, and what the tile's own
choice (SZL9_FORM 2, the product) gets: stage B's time per GiB, laboratory library.  python tools/lab/text_forms.py [MiB]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C, _lib
L = _lib.lab_lib(); _lib._lib = L
from sharpziplib_amd.batch import Engine
eng = Engine()
eng.debug_match_mode(0)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gens = {"mixed": lambda n: C.mixed(n, seed=5), "four_symbol": C.four_symbol, "period10": C.period10}
for kind, lv in (('logs', 9), ('logs', 6), ('enwik', 6), ('enwik', 9), ('dickens', 6), ('mixed', 6), ('four_symbol', 6), ('period10', 9)):
    d = gens[kind](mb << 20) if kind in gens else C.generate(kind, 0x106, 0, mb << 20)
    ref = None; out = []
    for form in (0, 1, 2):
        L.szl_debug_set(b"SZL9_FORM", form)
        best = 1e9
        for rep in range(3):
            r = eng.deflate([d], level=lv)[0]
            tm = eng.timing(); best = min(best, tm['match_ms'])
        if ref is None: ref = r.data
        out.append("form %d: B %.1f ms/GiB%s" % (form, best * 1024 / mb, "" if r.data == ref else " DIFFERENT BYTES"))
    print("%-12s L%d %4d MiB  %s" % (kind, lv, mb, "   ".join(out)), flush=True)
