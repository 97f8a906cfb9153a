"""Sharding of a cohort of samples over the GPUs of one node (one process per GPU).

`polee prep` fits every sample of an experiment independently (src/main.jl:560-660), so samples
are the unit of parallelism: each rank owns a contiguous block of samples and fits them on its own
GPU; no data-path collective is needed.  Only bookkeeping crosses ranks (barrier, max of the
elapsed time, sum of the iteration counts), through `torch.distributed` (backend "nccl" = RCCL on
the GPU box, "gloo" in the CPU tests)."""


def shard_samples(num_samples, world_size, rank):
    """Contiguous, balanced block of sample indices owned by `rank` (first ranks get the extras)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(num_samples, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def shard_rows(tcolptr, world_size, rank):
    """Contiguous block [r0, r1) of fragments (rows of X) owned by `rank` when ONE sample is spread over
    `world_size` GPUs, balanced on the number of non-zeros.  `tcolptr` = offsets of X in CSR form
    (the `xt=` input of RNASeqSample), 0- or 1-based.  Rows are locus-sorted, so a block touches a compact
    set of transcripts.  The likelihood and its gradient are sums over fragments: each rank evaluates its block
    and ONE all-reduce per pass (polee_comm) completes them."""
    import numpy as np
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    p = np.asarray(tcolptr).astype(np.int64)
    p = p - p[0]
    m, nnz = len(p) - 1, int(p[-1])
    cuts = [int(np.searchsorted(p, (nnz * r) // world_size, side="left")) for r in range(world_size + 1)]
    cuts[0], cuts[-1] = 0, m
    for r in range(1, world_size + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts[rank], cuts[rank + 1]


def take_rows(tcolptr, trowval, tnzval, r0, r1):
    """The CSR arrays of rows [r0, r1) with offsets rebased (same base as the input)."""
    import numpy as np
    p = np.asarray(tcolptr)
    base = p[0]
    lo, hi = int(p[r0] - base), int(p[r1] - base)
    return (p[r0:r1 + 1] - p[r0] + base).astype(p.dtype), np.asarray(trowval)[lo:hi], np.asarray(tnzval)[lo:hi]


def sample_seed(base_seed, sample_index):
    """Per-sample seed of the synthetic generator / device RNG: distinct streams per sample."""
    return (int(base_seed) + 7919 * int(sample_index)) & 0xFFFFFFFFFFFFFFFF


class Ranks:
    """Thin wrapper over torch.distributed that also works single-process."""

    def __init__(self, dist=None, device=None):
        self.dist, self.device = dist, device
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _reduce(self, value, op):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value):
        return self._reduce(value, self.dist.ReduceOp.MAX if self.dist is not None else None)

    def sum(self, value):
        return self._reduce(value, self.dist.ReduceOp.SUM if self.dist is not None else None)

    def gather(self, value):
        """Every rank's value, in rank order (a list of floats), on every rank."""
        if self.dist is None:
            return [float(value)]
        import torch
        t = torch.zeros(self.world, dtype=torch.float64, device=self.device or "cpu")
        t[self.rank] = float(value)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def aggregate_throughput(self, local_units, local_seconds):
        """Whole-job rate = units of all ranks / slowest rank's time."""
        return self.sum(local_units) / self.max(local_seconds)


def shard_regression_inputs(vars, x_init, F_arr, sample_scales, world_size, rank):
    """Inputs of RNASeqTranscriptLinearRegression for `rank` when the SAMPLES of the regression are sharded over the
    ranks (polee_regression_set_comm): this rank's rows of every per-sample array, and the column
    means of x_init over all samples (the shared parameters' initial values and the kernel-regression knots must be
    the same everywhere).  Per-sample arrays of `vars` are those with a leading dimension of S; a shared tree
    ([1, N] index arrays) is passed through.  Returns a dict of keyword arguments."""
    import numpy as np
    x_init = np.asarray(x_init)
    S = x_init.shape[0]
    rows = shard_samples(S, world_size, rank)
    if not rows:
        raise ValueError("rank %d would hold no sample (%d samples over %d ranks)" % (rank, S, world_size))
    sl = slice(rows[0], rows[-1] + 1)
    v = {k: (np.asarray(a)[sl] if np.ndim(a) >= 1 and np.shape(a)[0] == S and S > 1 else a) for k, a in vars.items()}
    return dict(vars=v, x_init=x_init[sl], F_arr=np.asarray(F_arr)[sl],
                sample_scales=np.asarray(sample_scales).reshape(S, -1)[sl],
                x_init_mean=x_init.astype(np.float64).mean(axis=0).astype(np.float32))


def approximate_likelihood_cohort(approx, samples, workers=2, device=0, on_result=None, **kwargs):
    """`approximate_likelihood` (likelihood-approximation.jl:395-624) for the samples of a cohort, `workers` of them in
    flight on ONE GPU.  Per sample the reference's `prep-sample` (main.jl:560-660) builds the tree from X on the CPU
    (hclust.jl), then fits; here a fit occupies the GPU for a fraction of a second while the host side of the same sample
    -- tree construction, device layout build -- takes seconds, so the samples are pipelined: every worker thread takes
    one sample through all stages on its own `Context` (= HIP stream); the C library releases the GIL, the host stages of
    some samples run under the device stage of others, and two fits that meet on the device share it (one's sparse pass
    under the other's tree kernels, like `bench.py --samples-per-gpu`).  Since round 4 the layout -- and with
    treemethod "cluster_device" / "cluster_auto" the tree -- is built on the GPU: the host side of a sample is loading and
    uploading, the cohort is bound by the GPU, and 2 - 4 workers are the measured optimum (C2-size samples: 3.2 - 3.4 samples/s
    with "cluster_auto", the tree on the host CPUs when they are idle and on the GPU otherwise -- the same tree either way;
    DESIGN 5.1).  With the host builders (POLEE_DEVICE_BUILD=0, treemethod "cluster" / "cluster_parallel") threads of one
    process share the address space the builders fill and release: two workers, or worker PROCESSES
    (`approximate_likelihood_cohort_processes`).

    samples: iterable of zero-argument callables, each returning `(m, n, colptr, rowval, nzval, effective_lengths)`
             or the dict of `h5io.read_likelihood_matrix` (the likelihood-matrix HDF5's arrays, rnaseq_sample.jl:505-519) -- called inside the worker, so that at most
             `workers` matrices are in host memory at a time -- or of such tuples.
    on_result(index, params): optional callback as results arrive (e.g. the prep HDF5 writer); else a list is returned.
    kwargs: passed to `approximate_likelihood` (num_steps, num_mc_samples, seed, gene_noninformative, ...).
    Returns the list of params dicts in the order of `samples` (None where `on_result` consumed them)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import core

    def job(item):
        idx, src = item
        lm = src() if callable(src) else src
        if isinstance(lm, dict):  # h5io.read_likelihood_matrix
            lm = tuple(lm[key] for key in ("m", "n", "colptr", "rowval", "nzval", "effective_lengths"))
        m, n, colptr, rowval, nzval, efflens = lm
        ctx = core.Context(device)
        sample, tree = core.sample_and_tree(approx, m, n, colptr, rowval, nzval, efflens, ctx=ctx)  # (side by side)
        params = core.approximate_likelihood(approx, sample, tree, **kwargs)
        del sample, tree
        if on_result is not None:
            on_result(idx, params)
            return None
        return params

    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as ex:
        return list(ex.map(job, enumerate(samples)))


def _process_init(host_threads, cache_mb=None, device_share=None, host_tree_slot=None):
    """device_share = (device, processes): this worker's cap on kept device buffers is its share of HALF the device's
    memory as the runtime reports it (hipMemGetInfo), at most 64 GiB (it was a constant sized for 288 GB)."""
    import os
    device_cache_mb = None
    if device_share is not None and "POLEE_DEVICE_CACHE_MB" not in os.environ:
        from . import core
        device, processes = device_share
        _, total = core.Context(device).mem_info()  # (asks the runtime; the block cache is not constructed before its first use)
        device_cache_mb = min(65536, (int(total) // 2 // max(1, int(processes))) >> 20)
    if host_tree_slot is not None:  # treemethod "cluster_auto": the slot for host-built trees is the cohort's, not the process's
        from . import core
        core._host_tree_slot = host_tree_slot
    if host_threads:
        os.environ["POLEE_HOST_THREADS"] = str(int(host_threads))  # (read once, when the library first needs it)
    if device_cache_mb is not None:
        # every worker process keeps its own freed DEVICE buffers (csrc/common.hpp, DevBlockCache): the GPU's memory is shared out
        os.environ.setdefault("POLEE_DEVICE_CACHE_MB", str(int(device_cache_mb)))
    if cache_mb is not None:
        # every worker process has its own scratch-block cache: the cohort's share of memory is divided between them
        # (four workers at the 8 GiB default pinned 32 GiB)
        from . import core
        core.host_cache_configure(int(cache_mb))


def _process_job(args):
    idx, src, treemethod, device, kwargs = args
    from . import core
    import numpy as np
    lm = src() if callable(src) else src
    if hasattr(lm, "keys"):  # the dict of h5io.read_likelihood_matrix, or an .npz of the same arrays
        lm = tuple(lm[key] for key in ("m", "n", "colptr", "rowval", "nzval", "effective_lengths"))
    m, n, colptr, rowval, nzval, efflens = lm
    m, n = int(np.ravel(m)[0]), int(np.ravel(n)[0])
    ctx = core.Context(device)
    approx = core.LogitSkewNormalPTTApprox(treemethod)
    sample, tree = core.sample_and_tree(approx, m, n, colptr, rowval, nzval, efflens, ctx=ctx)
    params = core.approximate_likelihood(approx, sample, tree, **kwargs)
    del sample, tree
    return idx, params


def usable_cpus():
    """CPUs this process may use: the affinity mask, cut to the cgroup's CFS quota when there is one (a container often
    shows every core of the host and throttles beyond its quota)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError):
            pass
    return n


def approximate_likelihood_cohort_processes(approx, samples, processes=4, host_threads=None, device=0, on_result=None,
                                            **kwargs):
    """As `approximate_likelihood_cohort`, with one worker PROCESS per sample in flight instead of a thread.  The host
    stages of a sample (tree construction, device layout build) fill and release gigabytes of memory; threads of one
    process share one address space, whose lock every page fault takes, so four samples in flight in one process
    prepare hardly more samples per second than one (measured on a 256-core host: 0.65 -> 0.76 samples/s).  Separate
    processes do not meet there; they share the GPU (every process its own HIP context, one fit = 1.2 GB).
    samples: picklable zero-argument callables (e.g. functools.partial(h5io.read_likelihood_matrix, path)) or tuples.
    host_threads: threads every process gives its builders (POLEE_HOST_THREADS; default: the usable CPUs -- affinity mask
    and cgroup quota -- divided by `processes`).  The pool is started with `spawn`.  Every worker keeps its builders'
    scratch blocks between samples (core.host_cache_configure): the single-process cap is divided by `processes`, so the
    cohort as a whole holds no more resident scratch than one process would."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    samples = list(samples)
    out = [None] * len(samples)
    mpctx = mp.get_context("spawn")
    slot = None
    if approx.treemethod == "cluster_auto":
        # one host-built tree at a time ACROSS the processes (it may then use every usable CPU: the layouts are built on the GPU)
        slot = mpctx.Semaphore(1)
        if host_threads is None:
            host_threads = usable_cpus()
    if host_threads is None:  # share the usable CPUs out (two host stages run side by side in every process)
        host_threads = max(2, usable_cpus() // max(1, int(processes)))
    from . import core
    cache_mb = max(256, core.host_cache_configure(-1) // max(1, int(processes)))
    with ProcessPoolExecutor(max_workers=max(1, int(processes)), mp_context=mpctx,
                             initializer=_process_init,
                             # (device buffers kept per process: half the GPU's memory shared out, asked of the runtime by the worker)
                             initargs=(host_threads, cache_mb, (device, int(processes)), slot)) as ex:
        jobs = [(i, s, approx.treemethod, device, kwargs) for i, s in enumerate(samples)]
        for idx, params in ex.map(_process_job, jobs):
            if on_result is not None:
                on_result(idx, params)
            else:
                out[idx] = params
    return out
