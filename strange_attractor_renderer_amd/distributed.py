"""Multi-GPU glue: one process per GPU, trajectories sharded by rank, Runtime::merge (reference
src/lib.rs:708-738) folded in rank order expressed as two collectives over xGMI.

The reference merges per-thread runtimes with a serial pairwise fold (src/lib.rs:1070-1076). Across GPUs the
same fold is order-free once written as reductions:
    count  : wrapping u32 add                      -> SUM over int32 (two's complement add wraps identically)
    zbuf   : max, ties -> the EARLIER runtime wins  -> MAX over int64 keys (sortable(z) << 32 | ~rank)
    steps  : payload of the zbuf winner            -> SUM over the two int32 halves of the f64 bits, every
                                                      rank but the winner contributing zeros (exact)
so the data path is ONE all-reduce(MAX, int64[npix]) + ONE reduce(SUM, int32[3*npix]) before colorize.
The device-side packing/unpacking lives behind the C ABI (sar_runtime_exchange_*); the collective is
torch.distributed's (backend "nccl" == RCCL on ROCm). PyTorch is plumbing here, nothing else.
"""
from __future__ import annotations


def shard_jobs(total_jobs: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous partition of jobs over ranks: rank r owns [first, first+count). Deterministic, covers every
    job exactly once, sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(total_jobs, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def exchange_merge(rt, rank: int, dist, key_buf, sum_buf, dst: int = 0):
    """Folds every rank's Runtime into rank `dst`'s (rank order == merge order). key_buf: int64[npix],
    sum_buf: int32[3*npix] torch tensors on the runtime's device. The pack/unpack kernels run on the RUNTIME's stream
    (a non-blocking stream of its own unless set), the collectives on torch's current stream: give the runtime that
    stream first — ``rt.set_stream(torch.cuda.current_stream().cuda_stream)`` — as bench.py does."""
    rt.exchange_export(rank, key_buf.data_ptr())
    dist.all_reduce(key_buf, op=dist.ReduceOp.MAX)
    rt.exchange_select(rank, key_buf.data_ptr(), sum_buf.data_ptr())
    dist.reduce(sum_buf, dst=dst, op=dist.ReduceOp.SUM)
    if rank == dst:
        rt.exchange_import(key_buf.data_ptr(), sum_buf.data_ptr())
