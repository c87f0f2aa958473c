"""gfxsim.harness — the product library on the interpreter.

build():   tools/gfxsim/_build/libfakehip.so (fakehip.cpp, g++), libszl_amd_sim.so (the product's own object files, the ones
           csrc/Makefile built for libszl_amd.so, linked against the fake runtime instead of libamdhip64) and the device assembly
           of every kernel translation unit (hipcc --cuda-device-only -S, same flags as the Makefile).
attach():  loads both, maps the arena into a gfxsim Runtime, loads the assembly, installs the launch callback and returns
           (runtime, ctypes library with the product's C ABI bound by sharpziplib_amd._lib).

With `use()` the Python mirrors of the package (batch.Engine, Deflater, Inflater, the stream classes) run against the simulated
device: the same host code, the same kernels' machine code, no GPU.  Test infrastructure only.
"""
import ctypes
import os
import subprocess
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sharpziplib_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
if os.path.join(ROOT, "tools") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tools"))

from gfxsim.runtime import Runtime, Memory, device_asm, SimError   # noqa: E402

KERNEL_UNITS = ["szl_kernels_match", "szl_kernels_match2", "szl_kernels_match9", "szl_kernels_fast", "szl_kernels_parse", "szl_kernels_block",
                "szl_kernels_checksum", "szl_kernels_inflate", "szl_kernels_inflate_exact", "szl_kernels_inflate_par", "szl_engine", "szl_api",
                "szl_api_inflate"]
_state = {}


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


LAB = bool(os.environ.get("GFXSIM_LAB"))       # the laboratory library (-DSZL_LAB=1: the dropped forms of stage B, SZL_SPEC_WB) instead of the product
if LAB:
    BUILD = os.path.join(BUILD, "lab")
    KERNEL_UNITS = KERNEL_UNITS + ["szl_kernels_match3", "szl_kernels_match5"]


def build(jobs=8):
    os.makedirs(BUILD, exist_ok=True)
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", CSRC, "libszl_amd_lab.so" if LAB else "libszl_amd.so"])
    fake = os.path.join(BUILD, "libfakehip.so")
    src = os.path.join(HERE, "fakehip.cpp")
    if _newer(fake, [src]):
        tmp = "%s.%d.tmp" % (fake, os.getpid())                     # (several suites may start at once: never a half-written library)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I/opt/rocm/include", src, "-o", tmp, "-lpthread"])
        os.replace(tmp, fake)
    objs = [os.path.join(CSRC, ("lab_" if LAB else "") + u + ".o") for u in KERNEL_UNITS]
    sim = os.path.join(BUILD, "libszl_amd_sim.so")
    if _newer(sim, objs + [fake]):
        tmp = "%s.%d.tmp" % (sim, os.getpid())
        subprocess.check_call(["g++", "-shared", "-fPIC", "-o", tmp, *objs, "-L" + BUILD, "-lfakehip", "-Wl,-rpath," + BUILD, "-lpthread"])
        os.replace(tmp, sim)
    # device assembly, in parallel
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".h", ".inc"))] + [os.path.join(ROOT, "include", "szl.h")]
    procs = []
    for u in KERNEL_UNITS:
        s, out = os.path.join(CSRC, u + ".hip"), os.path.join(BUILD, u + ".s")
        if _newer(out, [s] + hdrs):
            procs.append((u, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-DSZL_LAB=%d" % int(LAB),
                                               s, "-o", out], stderr=subprocess.DEVNULL)))
            if len(procs) >= jobs:
                u0, p0 = procs.pop(0)
                if p0.wait():
                    raise RuntimeError("device assembly of %s failed" % u0)
    for u0, p0 in procs:
        if p0.wait():
            raise RuntimeError("device assembly of %s failed" % u0)
    return sim


_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                       ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t)


def attach(verbose=False, fast_probe=False, memcheck=False, racecheck=False):
    """-> (Runtime, product C ABI through the simulated device).  Idempotent.
    fast_probe: k_probe_xchg_order (once per process, 512 rounds = 4.7 M instructions, 20 s here) runs 8 rounds instead — the
    interpreter serves an exchange in ascending lane order by construction, the probe has nothing to find out about it."""
    if "rt" in _state:
        return _state["rt"], _state["lib"]
    sim = build()
    fake = ctypes.CDLL(os.path.join(BUILD, "libfakehip.so"), mode=ctypes.RTLD_GLOBAL)
    fake.fakehip_arena_base.restype = ctypes.c_void_p
    fake.fakehip_arena_size.restype = ctypes.c_size_t
    fake.fakehip_alloc.restype = ctypes.c_void_p
    fake.fakehip_alloc.argtypes = [ctypes.c_size_t]
    base, size = fake.fakehip_arena_base(), fake.fakehip_arena_size()
    arena = np.ctypeslib.as_array((ctypes.c_ubyte * size).from_address(base))
    mem = Memory(buffer=arena, base=base)
    mem.alloc = lambda n, align=256: fake.fakehip_alloc(n)        # one allocator for host code and interpreter
    if memcheck:                                                  # every global access is checked against the runtime's shadow bytes
        fake.fakehip_shadow_base.restype = ctypes.c_void_p
        mem.shadow = np.ctypeslib.as_array((ctypes.c_ubyte * size).from_address(fake.fakehip_shadow_base()))
    rt = Runtime(mem)
    rt.verbose = verbose
    rt.racecheck = racecheck
    if os.environ.get("GFXSIM_SCHED"):         # schedule fuzzing: results must not depend on how the wavefronts interleave
        rt.sched = np.random.default_rng(int(os.environ["GFXSIM_SCHED"]))
    for u in KERNEL_UNITS:
        rt.load_file(os.path.join(BUILD, u + ".s"))
    errors = []
    import threading
    one_at_a_time = threading.Lock()          # the engine launches from several host threads (one per device): one interpreter, one launch at a time

    def on_launch(name, gx, gy, gz, bx, by, bz, args, shmem):
        with one_at_a_time:
            return _on_launch(name, gx, gy, gz, bx, by, bz, args, shmem)

    def _on_launch(name, gx, gy, gz, bx, by, bz, args, shmem):
        try:
            np.seterr(over="ignore", invalid="ignore", divide="ignore")     # (numpy's error state is per thread: the engine launches from worker threads too)
            kname = name.decode()
            mod, k = rt.kernels[kname]
            explicit = [a for a in k.args if not a.get(".value_kind", "").startswith("hidden_")]
            vals = []
            for i, a in enumerate(explicit):
                vals.append(ctypes.string_at(args[i], a[".size"]))
            if fast_probe and "k_probe_xchg_order" in kname:
                vals[0] = (8).to_bytes(4, "little")
            if rt.verbose:
                print("[gfxsim] %s grid=(%d,%d,%d) block=(%d,%d,%d) lds+%d" % (kname[:110], gx, gy, gz, bx, by, bz, shmem), flush=True)
            rt.launch(kname, (gx, gy, gz), (bx, by, bz), vals, dyn_lds=shmem)
            return 0
        except Exception as e:       # the C side turns this into hipErrorLaunchFailure; keep the reason
            errors.append("".join(traceback.format_exception_only(type(e), e)).strip())
            if rt.verbose:
                traceback.print_exc()
            return 1

    cb = _CB(on_launch)
    fake.fakehip_set_launch_callback(cb)
    from sharpziplib_amd import _lib as L
    lib = L._load(sim)
    _state.update(rt=rt, lib=lib, fake=fake, cb=cb, errors=errors)
    return rt, lib


def use(fast_probe=False, memcheck=False, racecheck=False):
    """route sharpziplib_amd's mirrors (Engine, Deflater, Inflater ...) through the simulated device.
    memcheck: global loads, stores and atomics are checked byte by byte — a read of device memory that no copy, memset or kernel
    store has written, or any access outside the allocations, is recorded with kernel and assembly line (Runtime.memcheck_report())."""
    rt, lib = attach(fast_probe=fast_probe, memcheck=memcheck, racecheck=racecheck)
    from sharpziplib_amd import _lib as L
    L._lib = lib
    return rt


def errors():
    return _state.get("errors", [])
