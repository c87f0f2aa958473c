cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/infprof; mkdir -p $O
for head in ${HEADS:-0}; do
  SZL_INF_FIND_HEAD_KIB=$head rocprofv3 --kernel-trace --stats --output-format csv -d $O/h$head -o k -- python $R/tools/gpu_inflate_big.py 1024 ${KIND:-enwik} > $O/h$head.log 2>&1
  f=$(find $O/h$head -name '*kernel_stats.csv' | head -1)
  echo "== head $head"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_find', 'k_convert', 'k_inflate', 'k_resolve')): print("  %-50s calls %s avg %.3f ms" % (n.split('(')[0][:50], r['Calls'], float(r['AverageNs']) / 1e6))
PY
  grep "MiB:" $O/h$head.log
done
