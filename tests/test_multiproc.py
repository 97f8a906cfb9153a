"""N>1 launch path of the cohort sharding (bench.py --gpus N) with world_size-2 gloo on CPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

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


ROW_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["POLEE_ROOT"])
import numpy as np, scipy.sparse as sp, torch
import torch.distributed as dist
from polee_amd.cohort import shard_rows, take_rows
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d = np.load(os.path.join(os.environ["POLEE_ROOT"], "tests", "golden", "mBr_M_6w_1.likelihood-matrix.npz"))
m, n = int(d["m"]), int(d["n"])
X = sp.csc_matrix((d["nzval"], d["rowval"].astype(np.int64) - 1, d["colptr"].astype(np.int64) - 1), shape=(m, n))
Xr = X.tocsr(); Xr.sort_indices()
tcolptr = (Xr.indptr + 1).astype(np.uint64); trowval = (Xr.indices + 1).astype(np.uint32); tnzval = Xr.data.astype(np.float32)
r0, r1 = shard_rows(tcolptr, world, rank)
bp, br, bv = take_rows(tcolptr, trowval, tnzval, r0, r1)
blk = sp.csr_matrix((bv, br.astype(np.int64) - 1, bp.astype(np.int64) - 1), shape=(r1 - r0, n)).tocsc()
blk.sort_indices()
so = O.Sample(r1 - r0, n, (blk.indptr + 1).astype(np.uint32), (blk.indices + 1).astype(np.uint32), blk.data.astype(np.float32))
x = np.random.default_rng(0).dirichlet(np.ones(n)).astype(np.float32)
lp, g = so.log_likelihood(x)
t = torch.tensor(np.concatenate([g, [lp, r1 - r0, blk.nnz]]))
dist.all_reduce(t)
if rank == 0:
    full = O.Sample(m, n, d["colptr"], d["rowval"], d["nzval"])
    lp_f, g_f = full.log_likelihood(x)
    tot = t.numpy()
    print(json.dumps({"rows": int(tot[-2]), "nnz": int(tot[-1]), "m": m, "nnz_full": int(X.nnz),
                      "lp_err": abs(tot[-3] - lp_f) / abs(lp_f), "g_err": float(np.abs(tot[:n] - g_f).max() / np.abs(g_f).max()),
                      "r": [int(r0), int(r1)]}))
dist.destroy_process_group()
'''


def test_shard_rows_is_a_balanced_partition():
    import numpy as np
    from polee_amd.cohort import shard_rows, take_rows
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 30, size=1000)
    p = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.uint64)  # 1-based offsets
    for W in (1, 2, 3, 8):
        cuts = [shard_rows(p, W, r) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 1000
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(W - 1))
        nnz = [int(p[b] - p[a]) for a, b in cuts]
        assert max(nnz) - min(nnz) <= 2 * 30  # balanced on non-zeros up to one row
    idx = np.arange(int(p[-1] - 1), dtype=np.uint32)
    bp, br, bv = take_rows(p, idx, idx.astype(np.float32), 10, 20)
    assert bp[0] == 1 and len(bp) == 11 and len(br) == int(p[20] - p[10]) and br[0] == int(p[10] - 1)


def test_two_rank_row_sharded_likelihood_sums_to_the_whole(tmp_path):
    """each rank evaluates its block of fragments (oracle on CPU here), one all-reduce of
    x_grad and lp gives the whole sample's values."""
    script = tmp_path / "row_worker.py"
    script.write_text(ROW_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, POLEE_ROOT=ROOT, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rows"] == d["m"] and d["nnz"] == d["nnz_full"]
    assert d["lp_err"] < 1e-12 and d["g_err"] < 1e-12


REG_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["POLEE_ROOT"])
import numpy as np, torch
import torch.distributed as dist
from polee_amd.cohort import shard_samples
from oracle import regression_ref as RR
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(7)          # the same problem on every rank
S, F, n, deg = 5, 2, 40, 4
x_init = rng.normal(-3, 1, size=(S, n))
design = np.zeros((S, F)); design[:, 0] = 1; design[2:, 1] = 1
ss = rng.normal(0, 0.2, size=S)
mean = x_init.mean(axis=0)
W = RR.kernel_regression_weights(1.0, mean, RR.choose_knots(mean.min(), mean.max(), deg))
vec = RR.flatten(RR.initial_params(x_init, F, deg), RR.PARAMS)
vec = vec + rng.normal(0, 0.2, size=vec.size)
p = RR.unflatten(vec, RR.PARAMS, S, F, n, deg)
eps = RR.unflatten(rng.normal(size=2 + 5 * F * n + 2 * n + S * n), RR.NOISE, S, F, n, deg)
rows = shard_samples(S, world, rank)
sl = slice(rows[0], rows[-1] + 1)
def cut(d, names):
    return {k: (v[sl] if k in names else v) for k, v in d.items()}
st, loss = RR.data_statistics(cut(p, ("qx_loc", "qx_softplus_scale")), cut(eps, ("x",)), design[sl], W, ss[sl], True, 0.9)
t = torch.tensor(np.concatenate([st.reshape(-1), [loss, len(rows)]]))
dist.all_reduce(t)                      # the step's one exchange
if rank == 0:
    tot = t.numpy()
    st_f, loss_f = RR.data_statistics(p, eps, design, W, ss, True, 0.9)
    kw = dict(W=W, x_bias_loc0=-2.0, x_bias_scale0=12.0, use_distortion=True, scale_penalty=0.9, use_point_estimates=False)
    whole, _ = RR.regression_loss(p, eps, design=design, sample_scales=ss, **kw)
    none = slice(0, 0)
    p0 = {k: (v[none] if k in ("qx_loc", "qx_softplus_scale") else v) for k, v in p.items()}
    e0 = {k: (v[none] if k == "x" else v) for k, v in eps.items()}
    prior, _ = RR.regression_loss(p0, e0, design=design[none], sample_scales=ss[none], **kw)
    print(json.dumps({"samples": int(tot[-1]), "S": S,
                      "stats_err": float(np.abs(tot[:-2] - st_f.reshape(-1)).max() / np.abs(st_f).max()),
                      "loss_err": abs(tot[-2] - loss_f) / abs(loss_f),
                      "decomposition_err": abs(prior + tot[-2] - whole) / abs(whole)}))
dist.destroy_process_group()
'''


def test_two_rank_sample_sharded_regression_statistics_sum_to_the_whole(tmp_path):
    """with the regression's samples sharded over ranks, one all-reduce of the (F+2) n observation
    statistics and the samples' loss terms reproduces the whole model (float64 restatement on CPU here; the device
    kernels are checked against the same decomposition in tests/test_gpu_regression.py)."""
    script = tmp_path / "reg_worker.py"
    script.write_text(REG_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, POLEE_ROOT=ROOT, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["samples"] == d["S"]
    assert d["stats_err"] < 1e-12 and d["loss_err"] < 1e-12 and d["decomposition_err"] < 1e-12


def test_shard_regression_inputs_partitions_the_samples():
    from polee_amd.cohort import shard_regression_inputs
    rng = np.random.default_rng(5)
    S, n, F = 5, 7, 2
    x_init = rng.normal(size=(S, n)).astype(np.float32)
    vars_ = dict(efflen=rng.normal(size=(S, n)), la_mu=rng.normal(size=(S, n - 1)), left_index=np.zeros((1, 13), np.int32))
    design, ss = rng.normal(size=(S, F)), rng.normal(size=(S, 1))
    parts = [shard_regression_inputs(vars_, x_init, design, ss, 3, r) for r in range(3)]
    np.testing.assert_array_equal(np.concatenate([p["x_init"] for p in parts]), x_init)
    np.testing.assert_array_equal(np.concatenate([p["F_arr"] for p in parts]), design)
    np.testing.assert_array_equal(np.concatenate([p["vars"]["la_mu"] for p in parts]), vars_["la_mu"])
    np.testing.assert_array_equal(np.concatenate([p["sample_scales"] for p in parts]), ss)
    for p in parts:
        assert p["vars"]["left_index"].shape == (1, 13)  # a shared tree is not cut
        np.testing.assert_allclose(p["x_init_mean"], x_init.mean(axis=0), rtol=1e-6)
    with pytest.raises(ValueError):
        shard_regression_inputs(vars_, x_init, design, ss, 8, 7)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_gpus_n_launches_n_ranks(world):
    """`python bench.py --gpus N` (no WORLD_SIZE in the environment) starts N ranks as a child torch.distributed.run
    and relays rank 0's JSON line as the last line of stdout.  No GPU here: POLEE_BENCH_DRY=1 stops after the
    rendezvous / barrier / max-over-ranks bookkeeping (the fits themselves are covered by the -m gpu twin in
    tests/test_gpu_multiproc.py)."""
    import json
    env = dict(os.environ, POLEE_BENCH_BACKEND="gloo", POLEE_BENCH_DRY="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for k in ("POLEE_HOST_THREADS",):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload", "c1",
                          "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    d = json.loads(last)
    assert d["n_gpus"] == world and d["ranks_seen"] == world and d["steps"] == 3 and d["warmup"] == 1
    assert abs(d["slowest_rank_s"] - 0.001 * world) < 1e-12  # the max over ranks, not rank 0's own time
    # the ranks share the CPUs this job may use (cgroup quota respected), not os.cpu_count() each
    from polee_amd.cohort import usable_cpus
    assert d["host_threads_per_rank"] == max(1, usable_cpus() // world)


def test_scale_script_launches_every_rank_count(tmp_path):
    """tools/scale.sh (one command for the first run on a multi-GPU node) through the same dry-run hook:
    1, 2 and 8 ranks of the weak and the row-sharded C2 launch, one summary line each, every rank seen."""
    env = dict(os.environ, POLEE_BENCH_BACKEND="gloo", POLEE_BENCH_DRY="1", OMP_NUM_THREADS="1", RANKS="1 2 8",
               WORKLOADS="weak rowshard", OUT=str(tmp_path), STEPS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "POLEE_HOST_THREADS"):
        env.pop(k, None)
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale.sh"), "8"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = open(os.path.join(str(tmp_path), "summary.txt")).read().strip().splitlines()
    assert len(lines) == 5, lines  # weak x {1, 2, 8} + row-sharded x {2, 8}
    for ln in lines:
        assert "FAILED" not in ln, ln
    assert "weak      ranks 8: dry run, 8 ranks seen" in lines and "rowshard  ranks 2: dry run, 2 ranks seen" in lines


def test_usable_cpus_is_positive_and_within_the_affinity_mask():
    import os
    from polee_amd.cohort import usable_cpus
    n = usable_cpus()
    assert 1 <= n <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
