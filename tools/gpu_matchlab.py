"""Stage-B laboratory (GPU box): sweeps the tuning knobs of the match kernels on one resident stream and prints
match_ms per configuration; every configuration's output must be byte-identical (sha256) to the first one's and, with
--oracle, to the oracle's.  Usage: python tools/gpu_matchlab.py [--mib 256] [--kind enwik] [--level 6] [--oracle] [--debug] cfg...
A cfg is NAME=V,NAME=V,... (knob names of include/szl.h szl_debug_set)."""
import argparse, hashlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from sharpziplib_amd import _lib, corpus
from sharpziplib_amd.batch import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=256)
ap.add_argument("--kind", default="enwik")
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--debug", action="store_true")
ap.add_argument("--product", action="store_true", help="the product library (one form of the full search) instead of the laboratory build")
ap.add_argument("cfgs", nargs="*")
a = ap.parse_args()
L = _lib.lib() if a.product else _lib.lab_lib()
_lib._lib = L                        # (everything below, Engine included, runs on that library)
n = a.mib << 20
seed = {"enwik": 0xE9, "logs": 0x106, "dickens": 0xD1CE}[a.kind]
host = corpus.generate(a.kind, seed, 0, n)
dev = torch.device("cuda", 0)
d_in = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_in[:n].copy_(torch.from_numpy(host))
eng = Engine()
streams, _, out_total = Engine.layout([n])
d_out = torch.empty(out_total + 64, dtype=torch.uint8, device=dev)
L.szl_engine_debug_match_mode(eng._h, 0)   # always the full search (the pilot would pick the on-demand form on logs)
ref_sha = None
if a.oracle:
    import oracle_ffi as O
    t = time.time(); ref_sha = hashlib.sha256(O.deflate(host, a.level)).hexdigest(); print("oracle %.1fs" % (time.time() - t), flush=True)
seen = set()
for cfg in (a.cfgs or ["SZL_MATCH_KERNEL=2"]):
    kv = dict(x.split("=") for x in cfg.split(","))
    for k in seen - set(kv):                       # knobs are sticky in the library: forget what this configuration does not name
        L.szl_debug_set(k.encode(), -2147483648)
    seen |= set(kv)
    for k, v in kv.items():
        L.szl_debug_set(k.encode(), int(v))
    L.szl_debug_set(b"SZL_DEBUG", 1 if a.debug else 0)
    ms = []
    for r in range(a.reps):
        eng.deflate_device(d_in.data_ptr(), d_out.data_ptr(), streams, level=a.level, flags=_lib.F_NOWRAP)
        tm = eng.timing(); ms.append(tm["match_ms"])
    sha = hashlib.sha256(d_out[:int(streams[0].out_len)].cpu().numpy().tobytes()).hexdigest()
    if ref_sha is None:
        ref_sha = sha
    print("%-70s match_ms %s  (per GiB %.1f) total %.1f links %.1f parse %.1f %s" % (
        cfg, " ".join("%.2f" % m for m in ms), min(ms) * 1024 / a.mib, tm["total_ms"], tm["links_ms"], tm["parse_ms"],
        "OK" if sha == ref_sha else "*** OUTPUT DIFFERS ***"), flush=True)
