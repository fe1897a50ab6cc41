"""bench.py's multi-rank control flow (barriers, rank-0-only sections, the data-parallel train step with its flat
all-reduce) with two ranks sharing the one visible GPU over gloo (GGD_BENCH_SHARE_GPU=1 test hook): it must finish and
print exactly one JSON line with the aggregate of both ranks.  The real 2/4/8-GPU runs use RCCL and one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu(native_lib):
    env = dict(os.environ, GGD_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
           "--warmup", "3", "--workload", "100k_512_cube", "--train-iters", "2", "--train-points", "20000",
           "--scenes-per-gpu", "2", "--no-decode", "--no-sweep", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0
    assert d["train"]["global_batch"] == 4 and d["train"]["allreduce_bytes"] > 0
    assert d["train_fused_decoder"]["iters_per_s"] > 0
