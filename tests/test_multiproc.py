"""N>1 launch path of the cohort sharding (bench.py --gpus N) with world_size-2 gloo on CPU."""
import os
import socket
import subprocess
import sys

from conftest import ROOT
from polee_amd.cohort import shard_samples, sample_seed

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["POLEE_ROOT"])
import torch.distributed as dist
from polee_amd.cohort import Ranks, shard_samples
dist.init_process_group("gloo")
r = Ranks(dist)
mine = shard_samples(5, r.world, r.rank)
r.barrier()
rate = r.aggregate_throughput(local_units=10 * len(mine), local_seconds=1.0 + r.rank)
tot = r.sum(len(mine))
if r.rank == 0:
    print(json.dumps({"world": r.world, "rate": rate, "total": tot, "mine": mine}))
dist.destroy_process_group()
'''


def test_shard_samples_is_a_balanced_partition():
    for S in (1, 6, 7, 64):
        for W in (1, 2, 4, 8):
            parts = [shard_samples(S, W, r) for r in range(W)]
            assert sorted(sum(parts, [])) == list(range(S))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
            assert all(p == list(range(p[0], p[0] + len(p))) for p in parts if p)
    assert sample_seed(1, 0) != sample_seed(1, 1)


def test_two_rank_gloo_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, POLEE_ROOT=ROOT, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    # 5 samples over 2 ranks: 3 + 2; whole-job rate = all units / slowest rank's time (2.0 s)
    assert d["world"] == 2 and d["total"] == 5 and d["mine"] == [0, 1, 2]
    assert abs(d["rate"] - 50 / 2.0) < 1e-9
