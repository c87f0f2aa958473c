"""Does stage B leave room for another call's stages A / C / D / E?  Two Engines, each with a HIP stream and a host thread of its own,
deflate one half of the headline's gigabyte each (two independent 512 MiB streams) — first one after the other, then at once, the second
thread started `lag` ms behind the first so that its stage B meets the first one's stages C-E.  Wall clock over both, device-resident.

    python tools/lab/two_engines_overlap.py [mib_each=512]
"""
import os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
from sharpziplib_amd import corpus, _lib
from sharpziplib_amd.batch import Engine

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = mib << 20
dev = torch.device("cuda", 0)
_lib.check(_lib.lib().szl_set_device(0), "szl_set_device")
flags = _lib.F_NOWRAP | _lib.F_CRC32
jobs = []
for k in range(2):
    host = corpus.generate("enwik", 0xE9, k * n, n)
    d_in = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_in[:n].copy_(torch.from_numpy(host))
    eng = Engine(); streams, _, out_total = Engine.layout([n])
    d_out = torch.empty(out_total + 64, dtype=torch.uint8, device=dev)
    jobs.append((eng, d_in, d_out, streams, torch.cuda.Stream(dev)))


def run(k):
    eng, d_in, d_out, streams, st = jobs[k]
    eng.deflate_device(d_in.data_ptr(), d_out.data_ptr(), streams, level=6, flags=flags, hip_stream=st.cuda_stream)
    st.synchronize()


def both(lag_ms):
    def second():
        _lib.lib().szl_set_device(0)
        if lag_ms: time.sleep(lag_ms / 1e3)
        run(1)
    t = threading.Thread(target=second); t0 = time.perf_counter(); t.start(); run(0); t.join()
    return (time.perf_counter() - t0) * 1e3


for k in range(2): run(k); run(k)
ref = [bytes(j[3][0].crc32.to_bytes(4, "little")) + int(j[3][0].out_len).to_bytes(8, "little") for j in jobs]
seq = []
for _ in range(4):
    t0 = time.perf_counter(); run(0); run(1); seq.append((time.perf_counter() - t0) * 1e3)
print("two %d MiB streams one after the other: %s ms" % (mib, " ".join("%.1f" % x for x in seq)), flush=True)
for lag in (0, 2, 5, 10, 15, 20):
    ts = [both(lag) for _ in range(4)]
    ok = ref == [bytes(j[3][0].crc32.to_bytes(4, "little")) + int(j[3][0].out_len).to_bytes(8, "little") for j in jobs]
    print("at once, the second %2d ms behind: %s ms  (same CRC-32 / length: %s)" % (lag, " ".join("%.1f" % x for x in ts), ok), flush=True)
t = jobs[0][0].timing(); print("engine 0's stages in the last run:", {k: round(v, 2) for k, v in t.items() if k.endswith("_ms")})
