"""The one cross-rank step of the path: the batch mean of log p(x).

Rows are independent in every coupling layer, so the batch shards over ranks with no
data-path collective: rank r evaluates its own rows, and a single all-reduce
of two float64 values -- (sum of log-prob, row count) -- produces the global mean.  With
backend "nccl" that is one RCCL all-reduce of 16 bytes over xGMI (latency-bound; ``async_op`` leaves it in
flight on RCCL's stream while the next batch is evaluated); the same code runs on "gloo" for the CPU
tests.  Activations never cross ranks.
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


_count_cache: dict = {}


def _count_on(device: torch.device, rows: int) -> torch.Tensor:
    """The row count as a one-element float64 device tensor, made once per (device, count): a host-to-device
    copy per evaluation would serialise the step behind it."""
    key = (device, int(rows))
    t = _count_cache.get(key)
    if t is None:
        if len(_count_cache) > 64:
            _count_cache.clear()
        t = _count_cache[key] = torch.tensor([float(rows)], dtype=torch.float64, device=device)
    return t


class PendingMean:
    """An all-reduce of (sum, count) in flight.  ``result()`` makes the current stream wait for it and returns
    the global mean; until then the stream is free, so the next evaluation's kernels run under the reduction
    (RCCL runs the collective on its own stream)."""

    def __init__(self, total: torch.Tensor, count: torch.Tensor, work, keep=None) -> None:
        self._total, self._count, self._work, self._keep = total, count, work, keep  # keep: the buffer in flight

    def result(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._total / self._count


def reduce_sum_count(local_sum: torch.Tensor, local_rows: int, group=None, async_op: bool = False,
                     force_collective: bool = False):
    """All-reduce (sum, count) as one 2-element float64 tensor on local_sum's device; returns (sum, count), or
    with ``async_op`` a ``PendingMean``.

    Single process: no device work at all (sum stays where it is, count is a host scalar tensor), so a
    caller's ``sum / count`` is one small kernel instead of four (``force_collective``: tests run the
    collective path on a one-rank group)."""
    multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
    if not multi:
        s, c = local_sum.reshape(-1)[0].to(torch.float64), torch.tensor(float(local_rows), dtype=torch.float64)
        return PendingMean(s, c, None) if async_op else (s, c)
    pair = torch.cat((local_sum.reshape(-1)[:1].to(torch.float64), _count_on(local_sum.device, local_rows)))
    work = None
    if pair.is_cuda and dist.get_backend(group) == "gloo":  # CPU-side test backend: reduce on the host
        host = pair.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        pair.copy_(host)
    elif async_op:
        work = dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group, async_op=True)
    else:
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
    return PendingMean(pair[0], pair[1], work, pair) if async_op else (pair[0], pair[1])


def sharded_mean_log_prob(
    log_prob_sum_fn: Callable[[torch.Tensor], torch.Tensor], x_local: torch.Tensor, group=None
) -> torch.Tensor:
    """Global mean of log p over all ranks' rows.

    `log_prob_sum_fn(x_local)` returns this rank's float64 sum of per-row log-probs (on the GPU:
    `model.log_prob(x, return_sum=True)[1]`, the HIP epilogue's sum)."""
    local = log_prob_sum_fn(x_local)
    total, count = reduce_sum_count(local, x_local.shape[0], group)
    return total / count


def sharded_mean_log_prob_async(
    log_prob_sum_fn: Callable[[torch.Tensor], torch.Tensor], x_local: torch.Tensor, group=None
) -> PendingMean:
    """The same with the reduction left in flight: call ``.result()`` when the mean is needed (e.g. after
    enqueueing the next batch), so the 16-byte all-reduce's latency is hidden behind the next evaluation."""
    return reduce_sum_count(log_prob_sum_fn(x_local), x_local.shape[0], group, async_op=True)


def global_column_mean(x_local: torch.Tensor, group=None) -> torch.Tensor:
    """Column means of the batch whose rows are sharded over the group's ranks: one all-reduce of the float64 vector
    [count, column sums].  Identical on every rank (an all-reduce hands every rank the same bits)."""
    buf = torch.cat([torch.tensor([float(x_local.shape[0])], dtype=torch.float64, device=x_local.device),
                     x_local.double().sum(dim=0)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[1:] / buf[0]


def global_column_std(x_local: torch.Tensor, group=None) -> torch.Tensor:
    """Unbiased column standard deviations (torch.std's default, what ActNormFlow's data-dependent initialisation uses:
    flows/affine_constant_flow.py:45) of the batch sharded over the group's ranks: the global mean first, then one
    all-reduce of [count, sum (x - mean)^2] in float64 (two passes: no cancellation)."""
    mean = global_column_mean(x_local, group)
    buf = torch.cat([torch.tensor([float(x_local.shape[0])], dtype=torch.float64, device=x_local.device),
                     ((x_local.double() - mean) ** 2).sum(dim=0)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return torch.sqrt(buf[1:] / (buf[0] - 1.0))
