"""More than one process on the GPU box: the launch path of `bench.py --gpus N` with real fits, and the exchange steps
of SURVEY.md 8(e) with TWO RANKS (row-sharded fit, sample-sharded regression).  With at least two GPUs they run over
RCCL; the driver's GPU boxes hold ONE MI355X, where RCCL cannot place two ranks: there both ranks share the GPU and the
all-reduce goes through the library's host-staged communicator (polee_comm_create_host) over a gloo group -- the same
polee_vi_set_comm / polee_regression_set_comm paths, a different transport."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _num_gpus():
    import torch
    return torch.cuda.device_count()


def _clean_env(**kw):
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _bench(args, **envkw):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_clean_env(**envkw),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_gpus_2_runs_two_fits_on_this_box():
    """Two ranks, each fitting its own C1 sample; with one GPU in the box both ranks share it (test hook) and only the
    bookkeeping collectives run, over gloo."""
    hooks = {} if _num_gpus() >= 2 else dict(POLEE_BENCH_BACKEND="gloo", POLEE_BENCH_FORCE_DEVICE="0")
    d = _bench(["--gpus", "2", "--workload", "c1", "--steps", "20", "--warmup", "3"], **hooks)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 20
    assert d["value"] > 0 and np.isfinite(d["value"])
    assert abs(d["value"] - 2 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]  # both ranks' iterations
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def _hooks():
    return {} if _num_gpus() >= 2 else dict(POLEE_BENCH_BACKEND="gloo", POLEE_BENCH_FORCE_DEVICE="0")


def test_bench_row_shard_two_ranks():
    d = _bench(["--gpus", "2", "--workload", "c1", "--steps", "20", "--warmup", "3", "--row-shard"], **_hooks())
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    # the first N > 1 run describes itself: determinism on by default, the build, every rank's kernel time, the exchange's time
    assert d["config"]["deterministic"] is True and "hipcc" in d["detail"]["build"]
    assert len(d["detail"]["kernel_ms_avg_per_rank"]) == 2 and all(v > 0 for v in d["detail"]["kernel_ms_avg_per_rank"])
    assert len(d["detail"]["allreduce"]["ms_avg_per_rank"]) == 2 and all(v > 0 for v in d["detail"]["allreduce"]["ms_avg_per_rank"])
    d2 = _bench(["--gpus", "2", "--workload", "c1", "--steps", "5", "--warmup", "1", "--row-shard", "--no-deterministic"], **_hooks())
    assert d2["config"]["deterministic"] is False


def test_bench_regression_two_ranks():
    d = _bench(["--gpus", "2", "--workload", "c3", "--steps", "20", "--warmup", "3"], **_hooks())
    assert d["n_gpus"] == 2 and d["detail"]["finite"]
    assert "hipcc" in d["detail"]["build"] and len(d["detail"]["allreduce"]["ms_avg_per_rank"]) == 2


ROW_FIT_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["POLEE_ROOT"])
import numpy as np, scipy.sparse as sp, torch
import torch.distributed as dist
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
host = os.environ.get("POLEE_TEST_TRANSPORT") == "host"  # both ranks on GPU 0, all-reduce staged through gloo
if host:
    local = 0
    dist.init_process_group("gloo")
else:
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
import polee_amd as P
from polee_amd.cohort import shard_rows, take_rows
g = os.path.join(os.environ["POLEE_ROOT"], "tests", "golden")
d = np.load(os.path.join(g, "mBr_M_6w_1.likelihood-matrix.npz")); pr = np.load(os.path.join(g, "mBr_M_6w_1.prep.npz"))
m, n = int(d["m"][0]), int(d["n"][0])
X = sp.csc_matrix((d["nzval"], d["rowval"].astype(np.int64) - 1, d["colptr"].astype(np.int64) - 1), shape=(m, n)).tocsr()
X.sort_indices()
tp, tr, tv = (X.indptr + 1).astype(np.uint64), (X.indices + 1).astype(np.uint32), X.data.astype(np.float32)
ctx = P.Context(local)
steps, K = 25, 6
z0 = np.random.default_rng(11).standard_normal(steps * K * (n - 1)).astype(np.float32)
def bcast(raw):
    box = [raw]; dist.broadcast_object_list(box, src=0); return box[0]
if host:
    comm = P.HostComm(ctx, world, rank, lambda a: dist.all_reduce(torch.from_numpy(a)))
else:
    comm = P.Comm(ctx, world, rank, broadcast=bcast)
r0, r1 = shard_rows(tp, world, rank)
s = P.RNASeqSample(r1 - r0, n, None, None, None, d["effective_lengths"], ctx=ctx, xt=take_rows(tp, tr, tv, r0, r1))
t = P.PolyaTreeTransform(pr["node_parent_idxs"], pr["node_js"], ctx=ctx)
fit = P.LikelihoodApproximationFit(s, t, num_steps=steps, num_mc_samples=K, z0=z0, comm=comm, gradonly=False)
fit.run(steps); fit.sync()
mine = np.concatenate(fit.params())
# ... and once more: a sample shared by several ranks is fitted with fixed-order gradient sums unless told otherwise (VERDICT r5
# item 5), so the second fit is the first one bit for bit
fit_b = P.LikelihoodApproximationFit(s, t, num_steps=steps, num_mc_samples=K, z0=z0, comm=comm, gradonly=False)
fit_b.run(steps); fit_b.sync()
again = np.concatenate(fit_b.params())
det_default = bool(fit.deterministic and fit_b.deterministic)
# the same fit on the whole sample, one rank's own stream, no communicator
s1 = P.RNASeqSample(m, n, None, None, None, d["effective_lengths"], ctx=ctx, xt=(tp, tr, tv))
f1 = P.LikelihoodApproximationFit(s1, t, num_steps=steps, num_mc_samples=K, z0=z0, gradonly=False)
f1.run(steps); f1.sync()
whole = np.concatenate(f1.params())
allp = [None] * world
dist.all_gather_object(allp, mine)
if rank == 0:
    print(json.dumps({"replica_diff": float(max(np.abs(a - allp[0]).max() for a in allp)),
                      "rerun_diff": float(np.abs(mine - again).max()), "deterministic_default": det_default,
                      "vs_whole": float(np.abs(mine - whole).max()), "scale": float(np.abs(whole).max()),
                      "lp_shard": float(fit.trace()[1][-1]), "lp_whole": float(f1.trace()[1][-1])}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("transport", ["rccl", "host"])
def test_two_rank_row_sharded_fit_equals_single_rank(tmp_path, transport):
    """The row-sharded VI path (all-reduce of g and lp per pass, polee_vi_set_comm) with TWO ranks: the fitted
    parameters are identical on both ranks and equal the single-rank fit of the whole sample up to the f32 summation
    order of the gradient.  "rccl": one GPU per rank; "host": both ranks on GPU 0, the library's host-staged
    communicator over gloo (runs on the 1-GPU boxes)."""
    if transport == "rccl" and _num_gpus() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has %d" % _num_gpus())
    import socket
    script = tmp_path / "row_fit_worker.py"
    script.write_text(ROW_FIT_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=_clean_env(POLEE_ROOT=ROOT, **({"POLEE_TEST_TRANSPORT": "host"} if transport == "host" else {})),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["replica_diff"] == 0.0
    assert d["deterministic_default"] and d["rerun_diff"] == 0.0  # two consecutive row-sharded fits: bitwise equal
    assert d["vs_whole"] < 2e-3 * d["scale"]
    assert abs(d["lp_shard"] - d["lp_whole"]) < 1e-5 * abs(d["lp_whole"])
