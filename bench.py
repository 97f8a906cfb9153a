#!/usr/bin/env python3
"""bench.py -- approx-lik VI iterations/sec on a synthetic RNA-Seq sample (BASELINE.json metric).

One "step" = one VI iteration of approximate_likelihood(::LogitSkewNormalPTTApprox)
(src/likelihood-approximation.jl:496-572): K=6 Monte-Carlo draws (sampling, Polya tree
transform, sparse log-likelihood + gradient over X, backward, ADAM).  Inputs (X in its device
layout, the tree, effective lengths) are resident in HBM before the timed region starts.

N=1 workload: BASELINE.json configs[1] -- one GENCODE-scale sample, n=200 000 transcripts x
m=30 000 000 fragments (~240 M nnz), on one MI355X.  N>1: one such sample per GPU, fitted
independently (samples shard one-per-GPU, no data-path collective: "scaling": "weak");
value = total VI iterations of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (sparse likelihood kernel vs HBM) and, at N=1,
`cpu_baseline` (the CPU oracle -- a port keeping the reference's loop structure -- timed on
this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (n, m, mean nnz/fragment)
    "c1": (1000, 100000, 2.2),
    # REAL-STRUCTURE workload: the reference's likelihood-matrix fixture (tests/golden) tiled block-diagonally 639 times
    "fixture": (313 * 639, 19743 * 639, 42775 / 19743),
    # ... at full size: every fragment of the tiled fixture nine times (seeded value factors): 113.5 M fragments, 246 M non-zeros
    "fixture_full": (313 * 639, 19743 * 639 * 9, 42775 / 19743),
    "small": (20000, 3000000, 8.0),
    "c2": (200000, 30000000, 8.0),
    "c5": (200000, 150000000, 8.0),
    # probe input of WIDE transcript sets (genes of ~20 isoforms: 87 % of the non-zeros in stream A2), C2's nnz
    "wide": (200000, 12000000, 20.0),
    # secondary metric: regression steps/s, S samples x F = 2 factors x n transcripts; own code path
    "c3": (200000, 6, 2),
    "c4": (200000, 8, 2),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_pass(nnz, m, n, K):
    """nnz*(4 B value + 4 B column) + (m+1)*4 B row offsets + K*n*8 B (read x, write grad)."""
    return nnz * 8 + (m + 1) * 4 + K * n * 8


def source_id():
    """Identifies the BUILD a measurement belongs to: a hash of the sparse kernel's sources and build flags (the GPU box
    has no .git).  A PMC capture (profiles/traffic_*.json) carries the id of the build it was taken with, and
    `roofline.traffic` is emitted only when it equals this run's."""
    import hashlib
    h = hashlib.sha1()
    for f in ("loglik.hip", "loglik_internal.hpp", "psell_build.cpp", "psell_device.hip", "psell_device.hpp", "wave.hpp", "common.hpp", "Makefile"):
        with open(os.path.join(ROOT, "polee_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def emit(out):
    """The ONE JSON line, as the last thing on stdout: native libraries (RCCL's version banner) write through C stdio,
    whose buffer would otherwise be flushed after Python's at exit."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks as a CHILD
    `python -m torch.distributed.run` (one process per GPU, RCCL) before this process has touched the GPU, relay the
    child's output and exit with its return code.  (No os.exec*: a process is never replaced here.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # host threads per rank: the CPUs this job may USE (affinity mask cut to the cgroup's CFS quota -- the GPU boxes show
    # 256 cores under a quota of 16) shared out over the ranks; never os.cpu_count()
    from polee_amd.cohort import usable_cpus
    per_rank = max(1, usable_cpus() // args.gpus)
    env.setdefault("OMP_NUM_THREADS", str(per_rank))
    env.setdefault("POLEE_HOST_THREADS", str(per_rank))  # (layout build threads per rank)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")  # re-emitted below as the LAST line
        else:
            sys.stdout.write(ln)
    rc = proc.wait()
    sys.stdout.flush()
    if line is not None:
        print(line, flush=True)
    sys.exit(rc if rc != 0 or line is not None else 1)


def init_ranks():
    """(world, rank, local_rank, dist-or-None, backend).  Test hooks (a box with fewer GPUs than ranks):
    POLEE_BENCH_BACKEND=gloo keeps the bookkeeping collectives on the CPU, POLEE_BENCH_FORCE_DEVICE=<id> puts every
    rank on that GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("POLEE_BENCH_BACKEND", "nccl")
    if "POLEE_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["POLEE_BENCH_FORCE_DEVICE"])
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
    return world, rank, local_rank, dist, backend


def make_comm(P, ctx, world, rank, dist, backend):
    """The fit's / model's communicator: RCCL (one GPU per rank), or -- test hook POLEE_BENCH_BACKEND=gloo, ranks sharing a
    GPU -- the library's host-staged communicator over the gloo group."""
    if dist is not None and backend != "nccl":
        import torch

        def allreduce(a):
            dist.all_reduce(torch.from_numpy(a))  # (shares the array's memory: summed in place)
        return P.HostComm(ctx, world, rank, allreduce)

    def bcast(raw):
        if dist is None:
            return raw
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    return P.Comm(ctx, world, rank, broadcast=bcast)


def regression_bench(args):
    """`--workload c3|c4`: the regression model's variational step (models/polee_regression.py fit): synthetic
    approximation parameters (mu ~ N(0,2), omega ~ N(-1,1), alpha ~ N(0,.3)), device RNG.  A step = one
    draw of every latent, loss + gradient, Adam.  C3 = BASELINE configs[2]: S = 6 samples in total, sharded over the
    ranks ("strong"); C4 = configs[3]: 8 samples per GPU ("weak", S = 8 x ranks).  With more than one rank the samples
    are sharded (cohort.shard_regression_inputs) and the (F+2) n sufficient statistics of the shared parameters are
    all-reduced once per step over RCCL (polee_regression_set_comm)."""
    import numpy as np
    world, rank, local_rank, dist, backend = init_ranks()
    import polee_amd as P
    from polee_amd.cohort import Ranks, shard_regression_inputs
    from tools import synth
    ranks = Ranks(dist, "cuda" if dist is not None and backend == "nccl" else None)
    n, S, F = WORKLOADS[args.workload]
    weak = args.workload == "c4"
    if weak:
        S *= world
    rng = np.random.default_rng(args.seed)  # every rank draws the whole cohort's inputs and keeps its rows
    smp = synth.make_sample(n, 1000000, 8.0, 1)
    parents, js = synth.make_tree(smp["gene"], 1)
    li, ri, fi = P.make_inverse_ptt_params(parents, js)
    ctx = P.Context(local_rank if world > 1 else 0)
    vars_ = dict(efflen=np.tile(smp["effective_lengths"], (S, 1)).astype(np.float32),
                 la_mu=rng.normal(0, 2, (S, n - 1)).astype(np.float32),
                 la_sigma=np.exp(rng.normal(-1, 1, (S, n - 1))).astype(np.float32),
                 la_alpha=rng.normal(0, .3, (S, n - 1)).astype(np.float32), left_index=li[None], right_index=ri[None],
                 leaf_index=fi[None])
    design = np.zeros((S, F), np.float32)
    design[:, 0] = 1
    design[S // 2:, 1] = 1
    # initial values: log of one draw of every sample's approximation (estimate.jl:436-455 uses the mean of 30)
    x0 = np.empty((S, n), np.float32)
    for s0 in range(0, S, 8):
        sl = slice(s0, min(S, s0 + 8))
        v = {k: (a[sl] if a.shape[0] == S and S > 1 else a) for k, a in vars_.items()}
        x0[sl] = np.log(np.maximum(P.RNASeqApproxLikelihood(v, ctx=ctx).sample(seed=1 + s0), 1e-12))
    scales = P.estimate_sample_scales(x0)
    comm = None
    if world > 1:
        comm = make_comm(P, ctx, world, rank, dist, backend)
        kw = shard_regression_inputs(vars_, x0, design, scales, world, rank)
        lik = P.RNASeqApproxLikelihood(kw["vars"], ctx=ctx)
        reg = P.RNASeqTranscriptLinearRegression(lik, kw["x_init"], kw["F_arr"], kw["sample_scales"], True, 1.0, False,
                                                 ctx=ctx, comm=comm, x_init_mean=kw["x_init_mean"])
    else:
        lik = P.RNASeqApproxLikelihood(vars_, ctx=ctx)
        reg = P.RNASeqTranscriptLinearRegression(lik, x0, design, scales, True, 1.0, False, ctx=ctx)

    def barrier():
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()
        ctx.synchronize()

    reg.fit(max(args.warmup, 1), seed=args.seed)
    barrier()
    t0 = time.perf_counter()
    trace = reg.fit(args.steps, seed=args.seed, return_trace=True)[-1]  # synchronises before returning
    barrier()
    elapsed = ranks.max(time.perf_counter() - t0)
    ar = None
    if comm is not None:  # (the first N > 1 run describes itself: the exchange's own HIP-event time on every rank)
        ar = {"count_f32": (F + 2) * n, "ms_avg_per_rank": ranks.gather(comm.allreduce_ms((F + 2) * n)),
              "what": "one all-reduce of (F+2) n f32 on the model's stream between two HIP events, mean of 20 calls behind an untimed first one"}
    if rank == 0:
        emit(({
            "metric": "regression steps/sec", "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: regression step, S=%d samples x F=%d factors x n=%d transcripts, 15 hinges, "
                                   "likelihood on, device RNG" % (args.workload.upper(), S, F, n),
                       "parallelism": "samples sharded over %d GPU(s), %d on rank 0, one all-reduce of (F+2) n f32 "
                                      "statistics per step" % (world, reg.num_samples) if world > 1 else "1 GPU",
                       "comm": comm.info() if comm is not None else None},
            "detail": {"build": P.version(), "allreduce": ar, "loss_first": float(trace[0]), "loss_last": float(trace[-1]),
                       "finite": bool(np.all(np.isfinite(trace))), "parameters": reg.num_params,
                       "note": "latency / gather bound (SURVEY.md 8(d)): no roofline claim for this step"},
        }))
    if dist is not None:
        dist.destroy_process_group()


def roofline_of(info, st0, st1, m, n, K, deterministic=False):
    """The `roofline` object of one measured input (stats of the fit before / after the timed steps).
    ALGORITHMIC bytes of a pass (CSR with u32 ids) and the bytes the device layout PHYSICALLY moves: the slice
    streams read once + the x windows read and the gradient windows added (dictionary entries x K x 4 B each way).  The
    layout stores a slice's transcript ids once (or as a 16-bit mask per fragment) and collapses single-transcript
    fragments, so physical < algorithmic and "algorithmic bytes / time" can exceed the HBM peak without the memory system
    being saturated: `frac` is therefore the PHYSICAL fraction -- bytes that really move / kernel time / peak -- and the
    contract's CSR-equivalent figure is reported beside it as `effective_*`.  The dominant kernel is the persistent launch
    over the sliced streams; the x-window gather, stream S's per-transcript kernel and (rare) the mixed stream's second
    launch are inside `pass_ms`."""
    launches = st1["loglik_kernel_launches"] - st0["loglik_kernel_launches"]

    def avg(key):
        return ((st1[key] * st1["loglik_kernel_launches"] - st0[key] * st0["loglik_kernel_launches"])
                / max(launches, 1))

    kern_ms, pass_ms = avg("loglik_kernel_ms_avg"), avg("loglik_pass_ms_avg")
    bytes_pass = algorithmic_bytes_per_pass(info["nnz"], m, n, K)
    sb = info["stream_bytes_hbm"]
    win_bytes = 2 * 4 * K * info.get("dict_entries", 0)
    phys_bytes_pass = sum(sb) + win_bytes
    uniform_share = sum(info["stream_nnz"][:5]) / max(info["nnz"], 1)  # (streams 0..4: the persistent launch)
    phys_bytes_dom = sum(sb[:5]) + win_bytes * (sum(info["stream_tiles"][:5]) / max(sum(info["stream_tiles"]), 1))
    bytes_dom = bytes_pass * uniform_share
    achieved = phys_bytes_dom / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    effective = bytes_dom / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        #  which of the two figures is which, in the line itself:
        "frac_definition": "physical: bytes the device layout really moves per launch / kernel time / peak.  SURVEY 8(d)'s "
                           "figure -- ALGORITHMIC (CSR-equivalent) bytes of the same rows / the same time / peak -- is "
                           "`effective_frac` (= frac_survey_8d)",
        "frac_survey_8d": effective / HBM_PEAK_GBS,
        "definition": "achieved = bytes the device layout moves per launch (slice streams + x / gradient windows) / "
                      "kernel time; effective_* = SURVEY 8(d)'s CSR-equivalent bytes of the same rows / the same time",
        "kernel": "loglik_stream_kernel<%d, false, false, %s>" % (K, "true" if deterministic else "false"),
        "kernel_ms_avg": kern_ms, "pass_ms_avg": pass_ms, "launches": int(launches),
        "physical_bytes_per_launch": phys_bytes_dom,
        "algorithmic_bytes_per_launch": bytes_dom,
        "effective_GBs": effective, "effective_frac": effective / HBM_PEAK_GBS,
        "physical_over_algorithmic": phys_bytes_dom / max(bytes_dom, 1),
        # the whole pass, same two definitions
        "pass_physical_GBs": phys_bytes_pass / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0,
        "pass_effective_GBs": bytes_pass / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0,
        "stream_share_of_nnz": [v / max(info["nnz"], 1) for v in info["stream_nnz"]],
        "stream_bytes_per_nnz": [b / max(v, 1) for b, v in zip(sb, info["stream_nnz"])],
        "layout_bytes_per_nnz": sum(sb) / max(info["nnz"], 1), "csr_bytes_per_nnz": (8 * info["nnz"] + 4 * (m + 1)) / max(info["nnz"], 1),
        "accumulate": "f32 row sums and gradient sums (MFMA + f32 atomics); the reference multiplies in f32 and accumulates in "
                      "f64 (sparse.jl:13-17,32-36): within 1e-4 of it at C2 / C5 (tests/test_gpu_configs.py)",
    }


def measure_other_input(P, synth, ctx, name, smp, args, K):
    """roofline.by_input: the same measurement (prewarm, warmup, timed steps; the fit's own HIP events) on another input of
    the same size class, after the headline's handles are gone.  Returns a compact dict."""
    parents, js = synth.make_tree(smp["gene"], seed=args.seed, kind=args.tree)
    m, n = smp["m"], smp["n"]
    sample = P.RNASeqSample(m, n, None, None, None, smp["effective_lengths"], ctx=ctx,
                            xt=(smp["tcolptr"], smp["trowval"], smp["tnzval"]))
    tree = P.PolyaTreeTransform(parents, js, ctx=ctx)
    if args.prewarm > 0:
        pre = P.LikelihoodApproximationFit(sample, tree, num_steps=args.prewarm, num_mc_samples=K, seed=args.seed + 1)
        pre.run(args.prewarm)
        pre.sync()
        del pre
    fit = P.LikelihoodApproximationFit(sample, tree, num_steps=args.warmup + args.steps, num_mc_samples=K, seed=args.seed,
                                       profile=max(1, args.profile_every), deterministic=args.deterministic)
    fit.run(args.warmup)
    fit.sync()
    st0 = fit.stats()
    ctx.synchronize()
    t0 = time.perf_counter()
    fit.run(args.steps)
    fit.sync()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    r = roofline_of(sample.info, st0, fit.stats(), m, n, K, args.deterministic)
    return {"input": name, "n": n, "m": m, "nnz": sample.info["nnz"], "value": args.steps / elapsed, "unit": "VI iters/s",
            "ms_per_step": 1e3 * elapsed / args.steps, "kernel_ms_avg": r["kernel_ms_avg"], "pass_ms_avg": r["pass_ms_avg"],
            "frac": r["frac"], "achieved": r["achieved"], "effective_frac": r["effective_frac"],
            "physical_bytes_per_launch": r["physical_bytes_per_launch"], "layout_bytes_per_nnz": r["layout_bytes_per_nnz"],
            "stream_share_of_nnz": r["stream_share_of_nnz"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("POLEE_BENCH_WORKLOAD", "c2"), choices=sorted(WORKLOADS))
    ap.add_argument("--draws", type=int, default=6, help="MC draws per VI iteration (LIKAP_NUM_MC_SAMPLES)")
    ap.add_argument("--tree", default="hclust", choices=["hclust", "balanced", "spine"])
    ap.add_argument("--cpu-steps", type=int, default=5, help="VI iterations timed for the CPU baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=123456789)
    ap.add_argument("--row-shard", action="store_true",
                    help="ONE sample for the whole job: its fragments are sharded over the GPUs and the likelihood "
                         "gradient is all-reduced once per pass (RCCL); strong scaling.  Default: one sample per GPU.")
    ap.add_argument("--deterministic", dest="deterministic", action="store_true", default=None,
                    help="fixed-order gradient sums (bitwise reproducible; polee_loglik_set_deterministic) instead of atomics. "
                         "Default: on for --row-shard with more than one rank (SURVEY 8(e): repeated N-rank runs of one sample are "
                         "bitwise stable), off otherwise")
    ap.add_argument("--no-deterministic", dest="deterministic", action="store_false",
                    help="float atomics even when the sample is row-sharded over several ranks")
    ap.add_argument("--prewarm", type=int, default=300,
                    help="VI iterations of a THROWAWAY fit of the same sample run before the warmup steps, so that the GPU "
                         "is at its sustained clocks when the short timed region starts (a real fit is 500 iterations; "
                         "the driver times 20); 0 = none")
    ap.add_argument("--set-diversity", type=float, default=0.0, metavar="P",
                    help="per-entry dropout probability of the generator (first entry of a fragment kept): fragments of a "
                         "gene stop sharing a handful of transcript sets; 0 = the generator as built")
    ap.add_argument("--generator", default="literal", choices=["literal", "patterns"],
                    help="set structure of the synthetic sample.  literal (default since round 4, the headline): every fragment "
                         "draws its OWN random non-empty subset of its gene's isoforms -- SURVEY 8(d)'s wording.  patterns: "
                         "the generator of rounds 1-3, a fragment draws one of its gene's <= 12 compatibility patterns (what "
                         "exon structure induces in real data: few distinct sets per gene); reported under roofline.by_input")
    ap.add_argument("--literal-subsets", action="store_true", help="(kept for scripts) same as --generator literal")
    ap.add_argument("--no-by-input", action="store_true",
                    help="skip the two extra inputs of roofline.by_input (the other generator and the tiled real fixture)")
    ap.add_argument("--profile-every", type=int, default=4, metavar="N",
                    help="bracket every N-th sparse pass of the timed region with HIP events (the roofline's kernel time); "
                         "bracketing every pass puts ~20 us of gaps into each C2 iteration (profiles/r05_event_gaps.txt)")
    ap.add_argument("--samples-per-gpu", type=int, default=1,
                    help="fits run concurrently on one GPU, each on its own stream (cohort mode; the headline uses 1)")
    args = ap.parse_args()
    if args.literal_subsets:
        args.generator = "literal"
    args.literal_subsets = args.generator == "literal"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)  # nothing above this line touches the GPU
    if args.workload in ("c3", "c4"):
        return regression_bench(args)

    world, rank, local_rank, dist, backend = init_ranks()
    if args.deterministic is None:  # (determinism is the multi-GPU default of a shared sample)
        args.deterministic = bool(args.row_shard and world > 1)
    if os.environ.get("POLEE_BENCH_DRY") == "1":
        # launch-path test hook for boxes without a GPU (tests/test_multiproc.py): rendezvous, barrier, max-over-ranks
        # and the JSON relay only -- no fit runs and no rate is reported
        from polee_amd.cohort import Ranks
        ranks = Ranks(dist, None)
        ranks.barrier()
        t, seen = ranks.max(0.001 * (rank + 1)), int(ranks.sum(1))
        if rank == 0:
            emit({"metric": "approx-lik VI iters/sec", "value": None, "unit": "VI iters/s", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "dry_run": True, "ranks_seen": seen,
                  "slowest_rank_s": t, "host_threads_per_rank": int(os.environ.get("POLEE_HOST_THREADS", "0"))})
        if dist is not None:
            dist.destroy_process_group()
        return

    import polee_amd as P
    from polee_amd.cohort import Ranks, sample_seed
    from tools import synth
    ranks = Ranks(dist, "cuda" if dist is not None and backend == "nccl" else None)

    n, m, mean_nnz = WORKLOADS[args.workload]
    K = args.draws
    S = max(1, args.samples_per_gpu)
    total = args.warmup + args.steps
    # every rank fits its own sample(s) (different seeds), as `polee prep` does over a cohort; with S > 1 the
    # fits of one GPU run concurrently, each on its own context (= HIP stream)
    t_gen = t_build = 0.0
    ctxs, fits, extra = [], [], []
    comm = None
    if args.row_shard:
        S = 1
    for si in range(S):
        t0 = time.time()
        # (row-sharded: every rank generates the same sample and keeps its block of fragments)
        if args.workload in ("fixture", "fixture_full"):
            smp_i = synth.tile_fixture(639, copies=9 if args.workload == "fixture_full" else 1)
        else:
            smp_i = synth.make_sample(n, m, mean_nnz, seed=sample_seed(args.seed, 0 if args.row_shard else rank * S + si),
                                      dropout=args.set_diversity, literal=args.literal_subsets)
        parents, js = synth.make_tree(smp_i["gene"], seed=args.seed, kind=args.tree)
        t_gen += time.time() - t0
        ctx_i = P.Context(local_rank if world > 1 else 0)
        t0 = time.time()
        xt_i, m_i = (smp_i["tcolptr"], smp_i["trowval"], smp_i["tnzval"]), m
        if args.row_shard:
            from polee_amd.cohort import shard_rows, take_rows
            r0, r1 = shard_rows(smp_i["tcolptr"], world, rank)
            xt_i, m_i = take_rows(smp_i["tcolptr"], smp_i["trowval"], smp_i["tnzval"], r0, r1), r1 - r0
            comm = make_comm(P, ctx_i, world, rank, dist, backend)
        sample_i = P.RNASeqSample(m_i, n, None, None, None, smp_i["effective_lengths"], ctx=ctx_i, xt=xt_i)
        tree_i = P.PolyaTreeTransform(parents, js, ctx=ctx_i)
        t_build += time.time() - t0
        fits.append(P.LikelihoodApproximationFit(sample_i, tree_i, num_steps=max(total, 1), num_mc_samples=K,
                                                 seed=args.seed, profile=max(1, args.profile_every), comm=comm, deterministic=args.deterministic))
        ctxs.append(ctx_i)
        if si == 0:
            smp, sample, tree, info = smp_i, sample_i, tree_i, sample_i.info
        else:
            extra.append((sample_i, tree_i))
    ctx, fit = ctxs[0], fits[0]

    def barrier():
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()

    if args.prewarm > 0:  # not part of the measurement: its own handle, destroyed before the warmup steps start
        pre = P.LikelihoodApproximationFit(sample, tree, num_steps=args.prewarm, num_mc_samples=K, seed=args.seed + 1,
                                           comm=comm)
        pre.run(args.prewarm)
        pre.sync()
        del pre
    for f in fits:
        f.run(args.warmup)
    for f in fits:
        f.sync()
    st0 = fit.stats()
    barrier()
    t_start = time.perf_counter()
    ctx.timer_start()
    for f in fits:
        f.run(args.steps)  # asynchronous: the S fits overlap on the device
    ev_ms = ctx.timer_stop()  # HIP events on the library's own stream (fit 0)
    for f in fits:
        f.sync()
    barrier()
    elapsed = time.perf_counter() - t_start
    st1 = fit.stats()

    elapsed = ranks.max(elapsed)  # slowest rank

    roof = roofline_of(info, st0, st1, m, n, K, args.deterministic)
    roof["event_sampling"] = ("every launch of the timed region is bracketed by HIP events" if args.profile_every <= 1 else
                              "every %d-th launch of the timed region is bracketed by HIP events on the library's stream (`launches` of "
                              "them): four event records per pass are ~20 us of stream gaps per C2 iteration" % args.profile_every)
    sid = source_id()
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, tools/profile.sh): a capture is
    # valid for the build and the workload it was taken with -- otherwise null
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tpath):
        try:
            cap = json.load(open(tpath))
            if ((cap.get("draws"), cap.get("tree"), cap.get("nnz"), cap.get("source_id")) == (K, args.tree, info["nnz"], sid)
                    and not args.deterministic):
                traffic = cap.get("hbm_bytes_per_launch")
                # where the figure comes from: a PMC capture of THIS build (source_id) on this workload, not of this run
                roof["traffic_source"] = {"file": "profiles/traffic_%s.json" % args.workload, "captured": cap.get("captured"),
                                          "commit": cap.get("commit"), "source_id": cap.get("source_id"),
                                          "pmc": cap.get("source")}
        except Exception:
            traffic = None
    roof["traffic"] = traffic
    input_name = ("tiled_real_fixture_x639" if args.workload == "fixture" else "tiled_real_fixture_x639_depth9" if args.workload == "fixture_full" else
                  ("literal_subsets" if args.literal_subsets else "gene_patterns") + ("_dropout_%g" % args.set_diversity if args.set_diversity else ""))

    out = {
        "metric": "approx-lik VI iters/sec",
        "value": (1 if args.row_shard else world * S) * args.steps / elapsed,
        "unit": "VI iters/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if args.row_shard else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s: one sample per GPU, n=%d transcripts x m=%d fragments, nnz=%d (%.2f/fragment), "
                        "K=%d draws per VI iteration, %s tree; set structure: %s" % (
                            args.workload.upper(), n, m, info["nnz"], info["nnz"] / m, K, args.tree,
                            "the reference's real fixture tiled x639" if args.workload == "fixture" else
                            "the reference's real fixture tiled x639, every fragment 9 times (seeded value factors)" if args.workload == "fixture_full" else
                            ("every fragment its own random subset of its gene's isoforms (SURVEY 8(d) literally)"
                             if args.literal_subsets else "one of <= 12 compatibility patterns per gene (generator of rounds 1-3)")
                            + (", per-entry dropout %g" % args.set_diversity if args.set_diversity else "")),
            "input": input_name, "comm": comm.info() if comm is not None else None,
            "samples_per_gpu": S, "draws": K, "tree": args.tree, "nnz": info["nnz"], "deterministic": bool(args.deterministic),
            "set_diversity": args.set_diversity, "literal_subsets": bool(args.literal_subsets),
            "parallelism": "one sample row-sharded over %d GPU(s), 1 all-reduce of K*n f32 per pass" % world
                           if args.row_shard else "sample-per-GPU x%d, no collective" % world if S == 1 else
                           "%d concurrent samples per GPU x%d GPUs, no collective" % (S, world),
        },
        "roofline": roof,
        "detail": {
            "build": P.version(),  # (compiler + loglik.hip tuning flags: a compiler bump that drops one costs up to 17 %)
            "source_id": sid, "hip_event_ms_per_step": ev_ms / args.steps, "prewarm_steps": args.prewarm, "gen_s": t_gen, "device_layout_build_s": t_build,
            "padded_nnz_ratio": info["padded_nnz"] / max(info["nnz"], 1), "num_tiles": info["num_tiles"],
            "max_tile_cols": info["max_tile_cols"],
        },
    }

    if world > 1:  # the first N > 1 run describes itself: every rank's kernel time, and the exchange's own time
        out["detail"]["kernel_ms_avg_per_rank"] = ranks.gather(roof["kernel_ms_avg"])
        out["detail"]["pass_ms_avg_per_rank"] = ranks.gather(roof["pass_ms_avg"])
        if comm is not None:
            ar = comm.allreduce_ms(n * K)
            out["detail"]["allreduce"] = {"count_f32": n * K, "ms_avg_per_rank": ranks.gather(ar), "what": "one all-reduce of K*n f32 on the fit's stream "
                                          "between two HIP events, mean of 20 calls behind an untimed first one", "comm": comm.info()}
    main_entry = {"input": input_name, "n": n, "m": m, "nnz": info["nnz"], "value": out["value"], "unit": "VI iters/s",
                  "ms_per_step": out["ms_per_step"], "kernel_ms_avg": roof["kernel_ms_avg"], "pass_ms_avg": roof["pass_ms_avg"],
                  "frac": roof["frac"], "achieved": roof["achieved"], "effective_frac": roof["effective_frac"],
                  "physical_bytes_per_launch": roof["physical_bytes_per_launch"],
                  "layout_bytes_per_nnz": roof["layout_bytes_per_nnz"], "stream_share_of_nnz": roof["stream_share_of_nnz"]}
    roof["by_input"] = {input_name: main_entry}

    cpu_inputs = None
    if rank == 0 and world == 1 and args.cpu_steps > 0 and S == 1:
        cpu_inputs = (smp, parents, js)
    by_input_wanted = (rank == 0 and world == 1 and S == 1 and not args.row_shard and not args.no_by_input
                       and args.workload == "c2" and not args.set_diversity)
    if by_input_wanted:
        # The same measurement on the inputs the headline's generator is not: the other generator at the same size, and
        # the reference's real fixture tiled to n = 200 k (real set structure: 54 % single-transcript fragments, 2.2
        # non-zeros per fragment).  The headline's device handles are released first.
        del fits, fit, sample, tree, extra, sample_i, tree_i, xt_i, smp_i  # (smp stays for the CPU baseline)
        import gc
        gc.collect()
        for name, make in (("gene_patterns" if args.literal_subsets else "literal_subsets",
                            lambda: synth.make_sample(n, m, mean_nnz, seed=sample_seed(args.seed, 0), literal=not args.literal_subsets)),
                           ("tiled_real_fixture_x639", lambda: synth.tile_fixture(639)),
                           # (round 5) the same set structure at C2's size: every fragment nine times, 246 M non-zeros
                           ("tiled_real_fixture_x639_depth9", lambda: synth.tile_fixture(639, copies=9))):
            try:
                smp_o = make()
                roof["by_input"][name] = measure_other_input(P, synth, ctx, name, smp_o, args, K)
                del smp_o
                gc.collect()
            except Exception as e:  # (never lose the headline line to a by-product)
                roof["by_input"][name] = {"input": name, "error": repr(e)}

    if cpu_inputs is not None:
        smp, parents, js = cpu_inputs
        # CPU baseline: the oracle (a port that keeps the reference's loop structure: CSR pass
        # threaded over fragments, CSC pass threaded over transcripts, serial tree walks) on the host cores.
        from oracle import oracle as O
        t0 = time.time()
        colptr, rowval, nzval = synth.to_csc(smp)
        so = O.Sample(m, n, colptr, rowval, nzval)
        to = O.PTT(parents, js)
        t_prep = time.time() - t0
        # the reference's default: one thread per PHYSICAL core (polee:8-12) -- of those this process may use (a container's
        # CPU quota counts: threads beyond it are only throttled)
        O.set_num_threads(O.physical_cores())
        t0 = time.time()
        O.approximate_likelihood(so, to, smp["effective_lengths"], num_steps=args.cpu_steps, num_mc=K,
                                 seed=args.seed)
        t_cpu = time.time() - t0
        out["cpu_baseline"] = {
            "value": args.cpu_steps / t_cpu, "unit": "VI iters/s", "cores": O.num_threads(), "kind": "port",
            "sample": "%d full VI iteration(s) (K=%d draws, %d likelihood passes) of the oracle on the SAME %s sample "
                      "(%.1f s of CPU work; setup %.1f s not counted); host CPUs usable: %s"
                      % (args.cpu_steps, K, 2 * K * args.cpu_steps, args.workload.upper(), t_cpu, t_prep,
                         "cgroup quota %d" % O.cpu_quota() if O.cpu_quota() else "all"),
        }
        out["detail"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]

    if rank == 0:
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
