#!/bin/bash
# The first box with two or more MI355X in it exercises what a one-GPU box cannot: one stream over several PHYSICAL devices (peer copies of
# the token gather, and the host-staged gather a box without peer access would take), the batch entry points over distinct ordinals, and
# bench.py's two launch shapes.  tools/gpu_tests.sh runs this when szl_device_count() >= 2; on one device it says so and returns 0.
#   usage: tools/gpu_two_device_check.sh        (logs under gpurun_out/two_device/)
set -u
cd "$(dirname "$0")/.."
n=$(python -c "from sharpziplib_amd import _lib; print(_lib.lib().szl_device_count())" 2>/dev/null || echo 0)
if [ "${n:-0}" -lt 2 ]; then echo "two-device check: $n device(s) — skipped (nothing has run between two physical devices yet, DESIGN.md section 6)"; exit 0; fi
out=gpurun_out/two_device; mkdir -p "$out"; rc=0
echo "two-device check on $n devices"
# 1. the multi-device tests pick distinct ordinals where the box has them (tests/test_gpu_multi.py: one stream over 2 / 3 / 5 slots, batches,
#    resident input) — peer copies between physical devices
python -m pytest tests/test_gpu_multi.py -q -m gpu > "$out/test_gpu_multi_peer.log" 2>&1 || rc=1
tail -n 2 "$out/test_gpu_multi_peer.log"
# 2. the same with the token gather forced through the host (what a box without peer access takes)
SZL_PART_HOST_GATHER=1 python -m pytest tests/test_gpu_multi.py -q -m gpu > "$out/test_gpu_multi_host_gather.log" 2>&1 || rc=1
tail -n 2 "$out/test_gpu_multi_host_gather.log"
# 3. bench.py as the driver launches it: one rank per GPU over RCCL (weak: a stream per GPU), and ONE stream over two devices (strong)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-extra-configs --no-cpu-baseline > "$out/bench_weak_2.json" 2> "$out/bench_weak_2.err" || rc=1
python bench.py --gpus 2 --mode strong --steps 5 --warmup 2 --no-extra-configs --no-cpu-baseline > "$out/bench_strong_2.json" 2> "$out/bench_strong_2.err" || rc=1
SZL_PART_HOST_GATHER=1 python bench.py --gpus 2 --mode strong --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline > "$out/bench_strong_2_host_gather.json" 2> "$out/bench_strong_2_host_gather.err" || rc=1
# 4. the multi-rank shapes of configs[2] and configs[4]: entries sharded over the ranks; log streams that change owner in an all-to-all over RCCL
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --workload cfg3 --steps 3 --warmup 1 > "$out/bench_cfg3_2.json" 2> "$out/bench_cfg3_2.err" || rc=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --workload cfg5 --steps 3 --warmup 1 > "$out/bench_cfg5_2.json" 2> "$out/bench_cfg5_2.err" || rc=1
for f in bench_weak_2 bench_strong_2 bench_strong_2_host_gather bench_cfg3_2 bench_cfg5_2; do python - "$out/$f.json" <<'PY' || rc=1
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
print("%-34s %10.1f %s  n_gpus %d  %.2f ms per step  parity: %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d.get("parity")))
PY
done
echo "two-device check rc=$rc (logs in $out/)"
exit $rc
