"""One runner for the one-off measurement scripts of tools/lab/ (GPU box, through `gpurun`):

    python tools/gpu_lab.py                      # the scripts and what each does
    python tools/gpu_lab.py <name> [args ...]    # run tools/lab/<name>.py with those arguments

They are bring-up, sweep and A/B helpers — streams above 2^31 bytes, data classes, the stage-B forms side by side, inflate chunk
sizes, one stream over several engines, `SetLevel` call patterns with a token-level diff — kept because DESIGN.md / DESIGN_HISTORY.md
quote their numbers; the scripts a round's evidence comes from stay one level up (gpu_profile_round.sh, gpu_matchlab.py,
gpu_stream_latency.py, gpu_small_call.py, gpu_fast.py, gpu_inflate_ab.py, gpu_inflate_big.py, gpu_tests.sh)."""
import ast
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LAB = os.path.join(HERE, "lab")


def main():
    names = sorted(f[:-3] for f in os.listdir(LAB) if f.endswith(".py"))
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help"):
        print(__doc__)
        for n in names:
            try:
                doc = ast.get_docstring(ast.parse(open(os.path.join(LAB, n + ".py")).read())) or ""
            except SyntaxError:
                doc = ""
            print("  %-18s %s" % (n, doc.strip().split("\n")[0][:150]))
        return 0
    name = sys.argv[1]
    if name not in names:
        print("unknown script %r (one of: %s)" % (name, ", ".join(names)))
        return 2
    os.chdir(ROOT)
    for p in (LAB, HERE, os.path.join(ROOT, "tests"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    path = os.path.join(LAB, name + ".py")
    sys.argv = [path] + sys.argv[2:]
    runpy.run_path(path, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
