import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # SZL_SIM=1: the GPU-marked tests run against tools/gfxsim (the product's machine code interpreted on the CPU) instead of a device —
    # `SZL_SIM=1 python -m pytest tests/test_gpu_gzip.py -m gpu`; only the small ones are practical (≈230 k wave-instructions per second)
    if os.environ.get("SZL_SIM"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from gfxsim import harness
        harness.use(fast_probe=True)


def pytest_collection_finish(session):
    # the interpreter suites (tests/test_sim_product_code.py) are a dozen single-threaded processes: started now, they run beside the
    # other CPU tests instead of after them
    if any("test_sim_product_code.py" in it.nodeid for it in session.items):
        try:
            import test_sim_product_code as T
            if os.path.exists(T.HIPCC):
                T.start_all()
        except Exception:
            pass                                       # (the module's own fixture reports what is wrong)


def pytest_sessionfinish(session, exitstatus):
    try:
        import test_sim_product_code as T
        T.stop_all()
    except Exception:
        pass


@pytest.fixture()
def lab_eng(monkeypatch):
    """An Engine on the LABORATORY library (libszl_amd_lab.so): the forms round 5 took out of the product library — the on-demand form of
    stage B and its pilot, k_match4, the second-walk emit — stay bit-exact there (tests/test_gpu_stage_b_forms.py has the other lab forms)."""
    from sharpziplib_amd import _lib
    from sharpziplib_amd.batch import Engine
    monkeypatch.setattr(_lib, "_lib", _lib.lab_lib())
    e = Engine()
    yield e
    e.debug_match_mode(-1)
    e.close()
