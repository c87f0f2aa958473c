"""One long text stream through the window pipeline, cold engine then warm: the stage times of each call (python tools/gpu_lab.py window_deflate_times [GiB=4])"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from sharpziplib_amd import corpus as C
from sharpziplib_amd.batch import Engine
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(gib * (1 << 30))
d = C.generate("enwik", 0x21B0, 0, n)
dev = torch.from_numpy(d).cuda()
eng = Engine()
st, _, ot = Engine.layout([n])
o = torch.empty(ot + 64, dtype=torch.uint8, device="cuda")
for rep in range(3):
    t = time.perf_counter()
    eng.deflate_device(dev.data_ptr(), o.data_ptr(), st, level=6, flags=3)
    wall = (time.perf_counter() - t) * 1e3
    tm = eng.timing()
    print("call %d: wall %.1f ms  events total %.1f  A %.1f  B %.1f  C %.1f  D %.1f  E %.1f  out %d" % (rep, wall, tm["total_ms"], tm["links_ms"], tm["match_ms"], tm["parse_ms"], tm["blocks_ms"], tm["encode_ms"], int(st[0].out_len)), flush=True)
