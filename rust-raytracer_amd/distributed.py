"""Multi-GPU sharding of the render: one process per GPU, sample-index ranges per rank, one sum.

Every camera path is independent and the film is a plain sum of per-sample contributions
(src/renderer.rs:199-204, src/film.rs:161-162, 228-229), so the work shards by sample index with
no exchange until the end, where the per-GPU accumulators are summed onto one rank: a single
reduce of float[h][w][4] (RCCL over xGMI when the process group's backend is "nccl").
"""


def sample_range(rank, world, total_samples):
    """Rank's share [begin, end) of sample indices 0..total_samples (strong scaling: the job is fixed)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return total_samples * rank // world, total_samples * (rank + 1) // world


def weak_sample_range(rank, samples_per_rank):
    """Rank's share when every rank renders the same number of samples (weak scaling)."""
    return rank * samples_per_rank, (rank + 1) * samples_per_rank


def reduce_film(film, dst=0, group=None):
    """Sums the per-rank film accumulators (a torch tensor [h][w][4]) onto rank `dst`, in place."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return film
