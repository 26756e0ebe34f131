"""CPU, world_size 2 over gloo: the multi-GPU exchange protocol (all-reduce MAX of depth keys + reduce SUM of
counts/steps halves) reproduces Runtime::merge folded in rank order, and job sharding covers every job once.

The pack/select/import arithmetic below is a numpy mirror of the three exchange kernels in
csrc/sar_image.hip (k_exch_export / k_exch_select / k_exch_import); the GPU versions are checked against
the same oracle merge in the -m gpu suite."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sortable(z32: np.ndarray) -> np.ndarray:
    b = (z32 + np.float32(0.0)).view(np.uint32)
    return np.where(b & 0x80000000, ~b, b | 0x80000000).astype(np.uint32)


def unsortable(s: np.ndarray) -> np.ndarray:
    b = np.where(s & 0x80000000, s & 0x7FFFFFFF, ~s).astype(np.uint32)
    return b.view(np.float32)


def exch_key(z32, rank):
    k = (sortable(z32).astype(np.uint64) << np.uint64(32)) | np.uint64(0xFFFFFFFF - rank)
    return (k ^ np.uint64(1 << 63)).view(np.int64)


def _worker(rank, world, port, W, H, jobs, n, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from strange_attractor_renderer_amd.distributed import shard_jobs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.poisson_saturne()
    cfg.width, cfg.height = W, H
    first, cnt = shard_jobs(jobs, world, rank)
    starts = O.start_points(seed, first, cnt)
    rt = O.Runtime(W, H)
    O.render_jobs(cfg, rt, starts, n)          # this rank's partial render (the oracle stands in for the GPU)
    npix = W * H
    # 1. export + all-reduce MAX
    key = torch.from_numpy(exch_key(rt.zbuf.ravel(), rank).copy())
    dist.all_reduce(key, op=dist.ReduceOp.MAX)
    red = key.numpy()
    # 2. select + reduce SUM (int32: count, then the two halves of the winner's steps bits)
    mine = exch_key(rt.zbuf.ravel(), rank) == red
    bits = np.where(mine, rt.steps.ravel().view(np.uint64), np.uint64(0))
    sums = np.empty(3 * npix, dtype=np.int32)
    sums[:npix] = rt.count.ravel().view(np.int32)
    sums[npix::2] = (bits & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    sums[npix + 1::2] = (bits >> np.uint64(32)).astype(np.uint32).view(np.int32)
    t = torch.from_numpy(sums)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        # 3. import
        s = t.numpy()
        count = s[:npix].view(np.uint32).reshape(H, W)
        k = (red.view(np.uint64) ^ np.uint64(1 << 63))
        zbuf = unsortable((k >> np.uint64(32)).astype(np.uint32)).reshape(H, W)
        sbits = s[npix::2].view(np.uint32).astype(np.uint64) | (s[npix + 1::2].view(np.uint32).astype(np.uint64) << np.uint64(32))
        steps = sbits.view(np.float64).reshape(H, W)
        q.put((count.copy(), zbuf.copy(), steps.copy(), max(rt.max, int(count.max()))))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_exchange_protocol_equals_merge_in_rank_order(oracle):
    from strange_attractor_renderer_amd.distributed import shard_jobs
    W, H, jobs, n, seed, world = 96, 80, 37, 4000, 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, jobs, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    count, zbuf, steps, mx = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expectation: per-rank partial renders merged with Runtime::merge, rank 0 first
    cfg = oracle.poisson_saturne()
    cfg.width, cfg.height = W, H
    parts = []
    for r in range(world):
        first, cnt = shard_jobs(jobs, world, r)
        rt = oracle.Runtime(W, H)
        oracle.render_jobs(cfg, rt, oracle.start_points(seed, first, cnt), n)
        parts.append(rt)
    acc = parts[0]
    for other in parts[1:]:
        assert oracle.merge(acc, other) == 0
    assert np.array_equal(count, acc.count)
    assert np.array_equal(zbuf.view(np.uint32), acc.zbuf.view(np.uint32))
    assert np.array_equal(steps.view(np.uint64), acc.steps.view(np.uint64))
    assert mx == acc.max
    # count (but not necessarily the tie-broken steps) is independent of the GPU count
    whole = oracle.Runtime(W, H)
    oracle.render_jobs(cfg, whole, oracle.start_points(seed, 0, jobs), n)
    assert np.array_equal(count, whole.count) and np.array_equal(zbuf.view(np.uint32), whole.zbuf.view(np.uint32))


def test_shard_jobs_partitions_exactly():
    from strange_attractor_renderer_amd.distributed import shard_jobs
    for total in (0, 1, 7, 64, 65536, 524288, 1000003):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            for r in range(world):
                first, cnt = shard_jobs(total, world, r)
                assert first == seen
                seen += cnt
            assert seen == total
    with pytest.raises(ValueError):
        shard_jobs(10, 2, 2)
