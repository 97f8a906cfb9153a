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

    def aggregate_throughput(self, local_units, local_seconds):
        """Whole-job rate = units of all ranks / slowest rank's time."""
        return self.sum(local_units) / self.max(local_seconds)
