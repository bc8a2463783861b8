"""Multi-GPU partition of one render: samples-per-pixel are split across ranks, each rank
accumulates its own full-frame buffer, and one all-reduce (sum) joins them.

The reference's only parallel driver is `threaded_render` (cpu_rt/src/ray_tracer.rs:479-535):
eight threads over disjoint vertical strips. Splitting sample indices instead keeps every GPU
on the whole frame (even load) and makes the union of the ranks' sample sets identical to the
single-GPU sample set, because a pixel-sample's random stream depends only on
(seed, pixel, sample index).
"""
from __future__ import annotations

from typing import Callable, Tuple


def sample_range(total_spp: int, rank: int, world_size: int) -> Tuple[int, int]:
    """First sample index and count for `rank`; ranges are contiguous, disjoint and cover total_spp."""
    if world_size < 1 or not (0 <= rank < world_size) or total_spp < 0:
        raise ValueError("bad rank / world_size / spp")
    base, rem = divmod(total_spp, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def render_sharded(render: Callable[[int, int], None], all_reduce_sum: Callable[[], None], total_spp: int,
                   rank: int, world_size: int, first_sample: int = 0) -> Tuple[int, int]:
    """Runs `render(first, count)` for this rank's share, then `all_reduce_sum()` once.

    `render` accumulates `count` samples starting at sample index `first` into the rank's
    buffer; `all_reduce_sum` sums the buffers of all ranks in place (RCCL over xGMI on GPUs:
    torch.distributed backend "nccl"; gloo in CPU tests).
    """
    first, count = sample_range(total_spp, rank, world_size)
    if count:
        render(first_sample + first, count)
    if world_size > 1:
        all_reduce_sum()
    return first_sample + first, count
