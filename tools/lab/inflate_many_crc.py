"""2048 x 4 MiB members inflated with and without their CRC-32 (device-resident, the bench's 4ii shape): python tools/gpu_lab.py inflate_many_crc [members=2048]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from sharpziplib_amd import _lib, corpus
if os.environ.get("SZL_AB_LIB"):
    _lib._lib = _lib._load(os.path.join(_lib.CSRC, os.environ["SZL_AB_LIB"]))
from sharpziplib_amd.batch import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msz = 4 << 20
eng = Engine()
plain = corpus.generate("enwik", 0xE9, 0, 1 << 30)
parts = [plain[i << 22:(i + 1) << 22] for i in range(256)]
comps = [r.data for r in eng.deflate(parts, level=6)]
comps = [comps[i % 256] for i in range(n)]
arr = (_lib.Stream * n)()
io = oo = 0
for i, b in enumerate(comps):
    arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_cap = io, len(b), oo, msz
    io += (len(b) + 3) & ~3; oo += msz
import numpy as np
hin = np.zeros(io + 8, np.uint8)
for s, b in zip(arr, comps):
    hin[s.in_off:s.in_off + s.in_len] = np.frombuffer(b, np.uint8)
d_in = torch.from_numpy(hin).cuda(); d_out = torch.empty(oo + 8, dtype=torch.uint8, device="cuda")
for flags, name in ((_lib.F_NOWRAP, "no checksum"), (_lib.F_NOWRAP | _lib.F_CRC32, "CRC-32")):
    best = 1e9; bw = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.inflate_device(d_in.data_ptr(), d_out.data_ptr(), arr, flags=flags)
        torch.cuda.synchronize(); bw = min(bw, (time.perf_counter() - t0) * 1e3)
        best = min(best, eng.timing()["inflate_ms"])
    assert all(s.status == 0 and s.out_len == msz for s in arr)
    print("%d x 4 MiB members, %-11s: inflate_ms %.1f  wall %.1f ms" % (n, name, best, bw), flush=True)
import zlib
assert arr[5].crc32 == zlib.crc32(parts[5].tobytes())
