"""bench.py's own launcher and line contract, on the CPU (no GPU, no library): `--gpus N` without a launcher must start N
ranks itself (torch.distributed.run on 127.0.0.1), run the barriers / MAX reduction / size gather, and print ONE JSON line
whose n_gpus is the number of ranks that ran.  The device work is replaced by a sleep (--stub, gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_gpus_2_launches_two_ranks_and_reports_them():
    p, lines = _run(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "0", "--mib", "4"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1                                    # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 0 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert rec["unit"] == "MiB/s" and rec["value"] > 0
    # MAX over ranks: rank 1 sleeps twice as long as rank 0 (2 ms vs 4 ms per step)
    assert rec["ms_per_step"] >= 3.9


def test_single_rank_needs_no_launcher():
    p, lines = _run(["--gpus", "1", "--stub", "--steps", "2", "--warmup", "0", "--mib", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


def test_world_size_that_contradicts_gpus_is_refused():
    """the driver starts the ranks itself and passes --gpus N: a line that says n_gpus != N must never be printed"""
    p, lines = _run(["--gpus", "4", "--stub", "--steps", "1", "--mib", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines and "WORLD_SIZE" in p.stderr


def test_cfg3_entries_shard_over_two_ranks_and_lay_out_one_archive():
    """--workload cfg3 (BASELINE configs[2]): contiguous groups of entries per rank, sizes all-gathered, the joint layout checked on rank 0"""
    p, lines = _run(["--gpus", "2", "--stub", "--workload", "cfg3", "--entries", "1001", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["steps"] == 2 and rec["unit"] == "MiB/s" and rec["value"] > 0
    assert rec["config"]["entries"] == 1001 and rec["config"]["parallelism"] == "entries-per-gpu x2"
    assert "roofline" in rec and "parity" in rec and rec["config"]["archive_payload_bytes"] > 0


def test_cfg5_streams_change_owner_in_one_all_to_all():
    """--workload cfg5 (BASELINE configs[4]): the skewed start, shard.rebalance's all-to-all (gloo here, nccl = RCCL on the GPUs), every
    stream compressed exactly once where it landed"""
    p, lines = _run(["--gpus", "2", "--stub", "--workload", "cfg5", "--total-mib", "1024", "--steps", "2", "--warmup", "0"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["streams_per_rank_before"] == [5, 11] and rec["config"]["streams_per_rank_after"] == [8, 8]
    assert "roofline" in rec and "parity" in rec and rec["rebalance_ms_rank0"] >= 0
