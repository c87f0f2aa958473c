"""config 5 (logs, level 9): the stage-B forms side by side on one resident stream (python tools/gpu_logs9.py [MiB])."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from sharpziplib_amd import corpus as C, _lib
from sharpziplib_amd.batch import Engine
L = _lib.lib(); eng = Engine()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for kind, lv in (('logs', 9), ('logs', 6), ('enwik', 9)):
    d = C.generate(kind, 0x106, 0, mb << 20)
    ref = None
    for mode in (0, 1, 2):
        eng.debug_match_mode(mode)
        for rep in range(2):
            r = eng.deflate([d], level=lv)[0]
        tm = eng.timing()
        used = eng.debug_match_mode()
        if ref is None: ref = r.data
        print(f"{kind} L{lv} {mb} MiB mode {mode} ({'on demand' if used else 'full'}): total {tm['total_ms']:.1f} ms, pilot {tm['pilot_ms']:.1f} B {tm['match_ms']:.1f} C {tm['parse_ms']:.1f} -> {mb/1024/(tm['total_ms']/1e3):.2f} GiB/s same={r.data == ref}", flush=True)
    eng.debug_match_mode(-1)
