"""Stage A's ticket pipeline (k_links3) relies on the LDS unit serving the lanes of one exchange in lane order — a property of
the hardware, probed once per device under load.  The product's tolerance is 0, so every call also re-derives the links of a few
sampled positions from first principles (k_links_guard) and, should one differ, distrusts the ticket form for the rest of the
process and runs the call again with the bucketed form (k_links2).  This test trips the guard on purpose (SZL_LINKS_GUARD_TEST)
and checks that the fallback produces the reference's bytes."""
import pytest

import oracle_ffi as O
from sharpziplib_amd import corpus as C

pytestmark = pytest.mark.gpu


def test_tripped_guard_reruns_the_call_with_the_bucketed_form(capfd):
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    L = _lib.lib()
    data = [C.generate("enwik", 41, 0, 5 << 20), C.generate("logs", 42, 0, 1 << 20)]
    eng = Engine()
    try:
        want = [O.deflate(d, 6) for d in data]
        assert [r.data for r in eng.deflate(data, level=6)] == want          # guard on (default), nothing to report
        L.szl_debug_set(b"SZL_LINKS_GUARD_TEST", 1)
        try:
            got = [r.data for r in eng.deflate(data, level=6)]
        finally:
            L.szl_debug_set(b"SZL_LINKS_GUARD_TEST", -2147483648)
        assert got == want
        assert "k_links2" in capfd.readouterr().err                           # it said so
        assert [r.data for r in eng.deflate(data, level=9)] == [O.deflate(d, 9) for d in data]   # and stays on the bucketed form
    finally:
        eng.close()
