"""Reads a rocprofv3 --kernel-trace csv of tools/lab/small_call_trace.py and prints the LAST call's kernels: start, duration, gap before."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by host work: the last long gap (> 100 us) starts the last call
starts = [i for i in range(1, len(rows)) if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 100000]
a = starts[-1] if starts else 0
t0 = int(rows[a]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("szl::", "")[:60]
    print("%8.1f us  +%6.1f gap  %7.1f us  grid %8s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], nm))
    prev = e; busy += e - s
print("kernels %d, device busy %.1f us of %.1f us" % (len(rows) - a, busy / 1e3, (prev - t0) / 1e3))
