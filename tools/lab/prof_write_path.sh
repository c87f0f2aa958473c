# kernel stats of one pipelined 1 GiB stream through the streaming Deflater (tools/lab/pipe_dbg.py):  bash tools/lab/prof_write_path.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wprof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/w -o k -- python $R/tools/lab/pipe_dbg.py 1024 > $O/w.log 2>&1
f=$(find $O/w -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("  %-56s calls %5s total %8.3f ms avg %8.3f ms" % (r['Name'].split('(')[0][:56], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6))
PY
grep "writes\|python:" $O/w.log
