"""The one cross-rank step of the path: the batch mean of log p(x).

Rows are independent in every coupling layer, so the batch shards over ranks with no
data-path collective: rank r evaluates its own rows, and a single all-reduce
of two float64 values -- (sum of log-prob, row count) -- produces the global mean.  With
backend "nccl" that is one RCCL all-reduce of 16 bytes over xGMI (latency-bound); the same
code runs on "gloo" for the CPU tests.  Activations never cross ranks.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def shard_bounds(total_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous row block [lo, hi) of rank `rank`; the first total % world ranks get one extra row."""
    base, extra = divmod(int(total_rows), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_sum_count(local_sum: torch.Tensor, local_rows: int, group=None) -> tuple[torch.Tensor, torch.Tensor]:
    """All-reduce (sum, count) as one 2-element float64 tensor on local_sum's device.

    Single process: no device work at all (sum stays where it is, count is a host scalar tensor), so a
    caller's ``sum / count`` is one small kernel instead of four."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if not multi:
        return local_sum.reshape(-1)[0].to(torch.float64), torch.tensor(float(local_rows), dtype=torch.float64)
    pair = torch.cat((local_sum.reshape(-1)[:1].to(torch.float64),
                      torch.tensor([float(local_rows)], dtype=torch.float64).to(local_sum.device, non_blocking=True)))
    if multi:
        if pair.is_cuda and dist.get_backend(group) == "gloo":  # CPU-side test backend: reduce on the host
            host = pair.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            pair.copy_(host)
        else:
            dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
    return pair[0], pair[1]


def sharded_mean_log_prob(
    log_prob_sum_fn: Callable[[torch.Tensor], torch.Tensor], x_local: torch.Tensor, group=None
) -> torch.Tensor:
    """Global mean of log p over all ranks' rows.

    `log_prob_sum_fn(x_local)` returns this rank's float64 sum of per-row log-probs (on the GPU:
    `model.log_prob(x, return_sum=True)[1]`, the HIP epilogue's sum)."""
    local = log_prob_sum_fn(x_local)
    total, count = reduce_sum_count(local, x_local.shape[0], group)
    return total / count
