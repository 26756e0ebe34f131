"""CPU, world_size 2 and 3 over gloo: strange_attractor_renderer_amd.distributed ITSELF — exchange_merge (rooted: all-reduce
MAX of depth keys + reduce SUM of counts / steps halves) and SlicedExchange / exchange_colorize (all-to-all of image
slices, merge in rank order, scalar all-reduce, sharded colorize, gather) — reproduces Runtime::merge folded in rank order
(reference src/lib.rs:708-738, 1068-1076), and job sharding covers every job once.

There is no HIP device here, so the Runtime and the exchange context handed to distributed.py are numpy stand-ins for the
exchange kernels of csrc/sar_image.hip (k_exch_export / _select / _import, k_exch_pack / _merge_slices / scalars, k_exch_flags /
_plan / _pack_sparse / _merge_sparse) working on the same raw buffers through the same pointers; the oracle stands in for the renderer. The process groups, the collectives, the
buffer geometry and the call sequence are the real ones; the HIP kernels run through the very same distributed.py
calls in tests/test_gpu_dist.py (-m gpu)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNSET = np.uint32(0x407FFFFF)  # sortable(-1.0f)


def sortable(z32: np.ndarray) -> np.ndarray:
    b = (z32 + np.float32(0.0)).view(np.uint32)
    return np.where(b & 0x80000000, ~b, b | 0x80000000).astype(np.uint32)


def unsortable(s: np.ndarray) -> np.ndarray:
    b = np.where(s & 0x80000000, s & 0x7FFFFFFF, ~s).astype(np.uint32)
    return b.view(np.float32)


def _at(ptr, dtype, n):
    return np.ctypeslib.as_array((ctypes.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(ptr)).view(dtype)


class NumpyRuntime:
    """The exchange half of api.Runtime with numpy in place of the HIP kernels (same buffers, same layouts)."""

    def __init__(self, ort, slice_pixels_fn):
        self.W, self.H = ort.width, ort.height
        self.npix = self.W * self.H
        self.count = ort.count.ravel().copy()
        self.z = sortable(ort.zbuf.ravel())
        self.steps = ort.steps.ravel().copy()
        self.max, self.wrap = ort.max, 0
        self.zmax = self.zmin = None
        self._slice_pixels = slice_pixels_fn

    def dims(self):
        return self.W, self.H

    def synchronize(self):
        pass

    # ---- rooted form (k_exch_export / k_exch_select / k_exch_import) ----
    def _key(self, rank):
        k = (self.z.astype(np.uint64) << np.uint64(32)) | np.uint64(0xFFFFFFFF - rank)
        return (k ^ np.uint64(1 << 63)).view(np.int64)

    def exchange_export(self, rank, key_ptr):
        _at(key_ptr, np.int64, self.npix)[:] = self._key(rank)

    def exchange_select(self, rank, key_ptr, sum_ptr):
        red = _at(key_ptr, np.int64, self.npix)
        out = _at(sum_ptr, np.int32, 3 * self.npix)
        mine = self._key(rank) == red
        bits = np.where(mine, self.steps.view(np.uint64), np.uint64(0))
        out[:self.npix] = self.count.view(np.int32)
        out[self.npix::2] = (bits & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
        out[self.npix + 1::2] = (bits >> np.uint64(32)).astype(np.uint32).view(np.int32)

    def exchange_import(self, key_ptr, sum_ptr):
        red = _at(key_ptr, np.int64, self.npix)
        s = _at(sum_ptr, np.int32, 3 * self.npix)
        self.count = s[:self.npix].view(np.uint32).copy()
        self.z = ((red.view(np.uint64) ^ np.uint64(1 << 63)) >> np.uint64(32)).astype(np.uint32)
        lo = s[self.npix::2].view(np.uint32).astype(np.uint64)
        hi = s[self.npix + 1::2].view(np.uint32).astype(np.uint64)
        self.steps = (lo | (hi << np.uint64(32))).view(np.float64).copy()
        self.max = max(self.max, int(self.count.max()))

    # ---- sliced form (k_exch_pack / k_exch_merge_slices / scalars) ----
    def exchange_pack(self, world, out_ptr):
        S = self._slice_pixels(self.npix, world)
        out = _at(out_ptr, np.uint8, world * S * 16)
        for d in range(world):
            blk = out[d * S * 16:(d + 1) * S * 16]
            lo, hi = min(self.npix, d * S), min(self.npix, (d + 1) * S)
            c, z, st = np.zeros(S, np.uint32), np.full(S, UNSET, np.uint32), np.zeros(S, np.float64)
            c[:hi - lo], z[:hi - lo], st[:hi - lo] = self.count[lo:hi], self.z[lo:hi], self.steps[lo:hi]
            blk[:S * 4] = c.view(np.uint8)
            blk[S * 4:S * 8] = z.view(np.uint8)
            blk[S * 8:] = st.view(np.uint8)

    def exchange_merge_slices(self, world, rank, in_ptr):
        S = self._slice_pixels(self.npix, world)
        buf = _at(in_ptr, np.uint8, world * S * 16)
        lo, hi = min(self.npix, rank * S), min(self.npix, (rank + 1) * S)
        n = hi - lo
        parts = []
        for r in range(world):
            blk = buf[r * S * 16:(r + 1) * S * 16]
            parts.append((blk[:S * 4].view(np.uint32)[:n].copy(), blk[S * 4:S * 8].view(np.uint32)[:n].copy(),
                          blk[S * 8:].view(np.float64)[:n].copy()))
        c, z, st = parts[0]
        if rank != 0:
            self.max, self.wrap = 0, 0
        for oc, oz, ost in parts[1:]:
            c = c + oc                                           # wrapping u32, :719
            if n:
                self.max = max(self.max, int(c.max()))           # running max over every intermediate sum, :721-723
            take = oz > z                                        # strict: the earlier rank wins ties, :728
            z = np.where(take, oz, z)
            st = np.where(take, ost, st)
        self.count[lo:hi], self.z[lo:hi], self.steps[lo:hi] = c, z, st
        seen = z[z != UNSET]
        self.zmax = max(int(sortable(np.float32([0.0]))[0]), int(seen.max()) if seen.size else 0)
        self.zmin = min(int(sortable(np.float32([np.finfo(np.float32).max]))[0]), int(seen.min()) if seen.size else 2**32 - 1)

    # ---- sparse form (k_exch_flags / k_exch_pack_sparse / k_exch_merge_sparse): records of the touched 64-pixel granules ----
    SEG = 64

    def _segments(self):
        nseg = (self.npix + self.SEG - 1) // self.SEG
        c, z, st = np.zeros(nseg * self.SEG, np.uint32), np.full(nseg * self.SEG, UNSET, np.uint32), np.zeros(nseg * self.SEG, np.float64)
        c[:self.npix], z[:self.npix], st[:self.npix] = self.count, self.z, self.steps
        return nseg, c.reshape(nseg, -1), z.reshape(nseg, -1), st.reshape(nseg, -1)

    def exchange_touched(self, ptr):
        nseg, c, z, _ = self._segments()
        _at(ptr, np.uint8, nseg)[:] = ((c != 0) | (z != UNSET)).any(axis=1)

    def exchange_pack_sparse(self, slot_ptr, out_ptr):
        nseg, c, z, st = self._segments()
        slot = _at(slot_ptr, np.int32, nseg)
        rec = self.SEG * 16
        for seg in np.nonzero(slot >= 0)[0]:
            blk = _at(out_ptr + int(slot[seg]) * rec, np.uint8, rec)
            blk[:self.SEG * 4] = c[seg].view(np.uint8)
            blk[self.SEG * 4:self.SEG * 8] = z[seg].view(np.uint8)
            blk[self.SEG * 8:] = st[seg].view(np.uint8)

    def exchange_merge_sparse(self, world, rank, slot_ptr, in_ptr):
        """A rank without a record holds the reset state (count 0, zbuf unset): materialise it and fold densely."""
        S = self._slice_pixels(self.npix, world)
        sps, rec = S // self.SEG, self.SEG * 16
        slot = _at(slot_ptr, np.int32, world * sps).reshape(world, sps)
        dense = np.zeros(world * S * 16, np.uint8)
        for r in range(world):
            c, z, st = np.zeros(S, np.uint32), np.full(S, UNSET, np.uint32), np.zeros(S, np.float64)
            for s in range(sps):
                if slot[r, s] >= 0:
                    blk = _at(in_ptr + int(slot[r, s]) * rec, np.uint8, rec)
                    c[s * self.SEG:(s + 1) * self.SEG] = blk[:self.SEG * 4].view(np.uint32)
                    z[s * self.SEG:(s + 1) * self.SEG] = blk[self.SEG * 4:self.SEG * 8].view(np.uint32)
                    st[s * self.SEG:(s + 1) * self.SEG] = blk[self.SEG * 8:].view(np.float64)
            blk = dense[r * S * 16:(r + 1) * S * 16]
            blk[:S * 4], blk[S * 4:S * 8], blk[S * 8:] = c.view(np.uint8), z.view(np.uint8), st.view(np.uint8)
        self.exchange_merge_slices(world, rank, dense.ctypes.data)

    def exchange_scalars_export(self, ptr):
        _at(ptr, np.int64, 4)[:] = [self.max, self.wrap, self.zmax, (~np.uint32(self.zmin)) & 0xFFFFFFFF]

    def exchange_scalars_import(self, ptr):
        v = _at(ptr, np.int64, 4)
        self.max, self.wrap, self.zmax = int(v[0]), int(v[1]), int(v[2])
        self.zmin = int(~np.uint32(v[3]) & 0xFFFFFFFF)


class NumpyExchange:
    """api.Exchange (the library's context object, csrc/sar_exchange.cpp) over a NumpyRuntime: the same geometry, the plan of the
    sparse form (k_exch_plan: where each of my records goes, where each record I receive arrives, the split sizes), the choice
    between sparse and dense, and the kernels' stand-ins above."""

    def __init__(self, rt, world, rank):
        self.runtime, self.world, self.rank = rt, world, rank
        self.slice_pixels = rt._slice_pixels(rt.npix, world)
        self.first = min(rt.npix, rank * self.slice_pixels)
        self.count = min(rt.npix, self.first + self.slice_pixels) - self.first
        self.granules = (rt.npix + rt.SEG - 1) // rt.SEG
        self.block_bytes = world * self.slice_pixels * 16
        self._sparse = False

    def flags(self, ptr):
        self.runtime.exchange_touched(ptr)

    def pack(self, flags_all_ptr, dense_above, send_ptr):
        world, rank, nseg, sps = self.world, self.rank, self.granules, self.slice_pixels // self.runtime.SEG
        sparse = False
        if flags_all_ptr:
            fa = _at(flags_all_ptr, np.uint8, world * nseg).reshape(world, nseg) != 0
            sparse = int(fa.sum()) <= dense_above * world * nseg
        self._sparse = sparse
        if not sparse:
            self.runtime.exchange_pack(world, send_ptr)
            return False, [self.slice_pixels * 16] * world, [self.slice_pixels * 16] * world
        fp = np.zeros((world, world * sps), bool)
        fp[:, :nseg] = fa
        mine = fp[rank]
        send_slot = np.where(mine, np.cumsum(mine) - 1, -1).astype(np.int32)[:nseg]
        sub = fp[:, rank * sps:(rank + 1) * sps].reshape(-1)
        self._recv_slot = np.where(sub, np.cumsum(sub) - 1, -1).astype(np.int32)
        self.runtime.exchange_pack_sparse(np.ascontiguousarray(send_slot).ctypes.data, send_ptr)
        rec = self.runtime.SEG * 16
        return True, [int(c) * rec for c in mine.reshape(world, sps).sum(axis=1)], [int(c) * rec for c in sub.reshape(world, sps).sum(axis=1)]

    def merge(self, recv_ptr, scalars_ptr):
        if self._sparse:
            self.runtime.exchange_merge_sparse(self.world, self.rank, self._recv_slot.ctypes.data, recv_ptr)
        else:
            self.runtime.exchange_merge_slices(self.world, self.rank, recv_ptr)
        self.runtime.exchange_scalars_export(scalars_ptr)

    def finish(self, scalars_ptr):
        self.runtime.exchange_scalars_import(scalars_ptr)

    def rooted(self, step, key_ptr, sum_ptr=0):
        if step == 0:
            self.runtime.exchange_export(self.rank, key_ptr)
        elif step == 1:
            self.runtime.exchange_select(self.rank, key_ptr, sum_ptr)
        else:
            self.runtime.exchange_import(key_ptr, sum_ptr)


class NumpyApi:
    """The api.* names SlicedExchange uses, for a NumpyRuntime: the exchange context above (slice geometry from the real library, a
    host function), colorize through the oracle with the GLOBAL max."""

    Exchange = NumpyExchange

    def __init__(self, O, S):
        self.O, self.S = O, S

    def exchange_slice_pixels(self, npix, world):
        return self.S.exchange_slice_pixels(npix, world)

    def colorize_range_device(self, cfg, rt, first, n, out_ptr):
        O = self.O
        ort = O.Runtime(rt.W, rt.H)
        ort.count[:] = rt.count.reshape(rt.H, rt.W)
        ort.steps[:] = rt.steps.reshape(rt.H, rt.W)
        ort.zbuf[:] = unsortable(rt.z).reshape(rt.H, rt.W)
        ort.set_max(0xFFFFFFFF if rt.wrap else rt.max)
        img = O.colorize(cfg, ort).reshape(-1, 4)
        _at(out_ptr, np.uint16, n * 4)[:] = img[first:first + n].ravel()


def _worker(rank, world, port, W, H, jobs, n, seed, mode, q, scale=1.0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import strange_attractor_renderer_amd as S
    from strange_attractor_renderer_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.poisson_saturne()
    cfg.width, cfg.height, cfg.transparent, cfg.scale = W, H, 0, scale
    first, cnt = D.shard_jobs(jobs, world, rank)
    ort = O.Runtime(W, H)
    O.render_jobs(cfg, ort, O.start_points(seed, first, cnt), n)   # this rank's partial render (the oracle stands in for the GPU)
    rt = NumpyRuntime(ort, S.exchange_slice_pixels)
    npix = W * H
    if mode == "rooted":
        key = torch.empty(npix, dtype=torch.int64)
        sums = torch.empty(3 * npix, dtype=torch.int32)
        D.exchange_merge(NumpyExchange(rt, world, rank), dist, key, sums, dst=0)   # the real function, real collectives
        if rank == 0:
            q.put((rt.count.reshape(H, W).copy(), unsortable(rt.z).reshape(H, W).copy(), rt.steps.reshape(H, W).copy(), rt.max, None))
    else:
        # "sliced": whole slices; "sparse": records of the touched segments, whatever share of the image they are;
        # "auto": sparse unless more than half of the segments are touched
        ex = D.SlicedExchange(NumpyApi(O, S), cfg, rt, rank, world, "cpu", sparse=mode != "sliced",
                              dense_above=2.0 if mode == "sparse" else 0.5)
        img = D.exchange_colorize(ex, dist, dst=0)                  # pack, all-to-all, merge, scalars, colorize, gather
        f, c = ex.first, ex.count
        q.put((rank, f, c, rt.count[f:f + c].copy(), unsortable(rt.z[f:f + c]).copy(), rt.steps[f:f + c].copy(), rt.max,
               img.numpy().view(np.uint16).reshape(H, W, 4).copy() if rank == 0 else None, ex.bytes_on_the_wire()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(world, mode, W, H, jobs, n, seed, n_results, scale=1.0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, jobs, n, seed, mode, q, scale)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(n_results)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _fold_in_rank_order(oracle, W, H, jobs, n, seed, world, scale=1.0):
    from strange_attractor_renderer_amd.distributed import shard_jobs
    cfg = oracle.poisson_saturne()
    cfg.width, cfg.height, cfg.transparent, cfg.scale = W, H, 0, scale
    parts = []
    for r in range(world):
        first, cnt = shard_jobs(jobs, world, r)
        rt = oracle.Runtime(W, H)
        oracle.render_jobs(cfg, rt, oracle.start_points(seed, first, cnt), n)
        parts.append(rt)
    acc = parts[0]
    for other in parts[1:]:
        assert oracle.merge(acc, other) == 0
    return cfg, acc


@pytest.mark.timeout(300)
def test_exchange_merge_equals_merge_in_rank_order(oracle):
    W, H, jobs, n, seed, world = 96, 80, 37, 4000, 5, 2
    (count, zbuf, steps, mx, _), = _spawn(world, "rooted", W, H, jobs, n, seed, 1)
    cfg, acc = _fold_in_rank_order(oracle, W, H, jobs, n, seed, world)
    assert np.array_equal(count, acc.count)
    assert np.array_equal(zbuf.view(np.uint32), acc.zbuf.view(np.uint32))
    assert np.array_equal(steps.view(np.uint64), acc.steps.view(np.uint64))
    assert mx == acc.max
    # count (but not necessarily the tie-broken steps) is independent of the GPU count
    whole = oracle.Runtime(W, H)
    oracle.render_jobs(cfg, whole, oracle.start_points(seed, 0, jobs), n)
    assert np.array_equal(count, whole.count) and np.array_equal(zbuf.view(np.uint32), whole.zbuf.view(np.uint32))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,mode,W,H,scale", [(2, "sliced", 97, 83, 1.0), (3, "sliced", 97, 83, 1.0), (2, "sparse", 97, 83, 1.0),
                                                  (3, "sparse", 97, 83, 1.0), (3, "auto", 150, 420, 0.3), (2, "auto", 97, 83, 1.0)])
def test_sliced_exchange_colorize_equals_merge_in_rank_order(oracle, world, mode, W, H, scale):
    """Dense slices, sparse records (forced, and chosen by the share of touched segments: a view that fills a tenth of a tall
    image leaves most segments untouched; one that covers the image goes dense) — the same merged frame."""
    jobs, n, seed = 41, 3000, 6                                    # npix % world != 0, jobs % world != 0
    got = _spawn(world, mode, W, H, jobs, n, seed, world, scale)
    cfg, acc = _fold_in_rank_order(oracle, W, H, jobs, n, seed, world, scale)
    count, zbuf, steps = np.zeros(W * H, np.uint32), np.zeros(W * H, np.float32), np.zeros(W * H)
    img, covered = None, 0
    wire = {r[0]: r[8] for r in got}
    got = [r[:8] for r in got]
    if mode == "sliced":
        assert all(w["form"] == "dense" for w in wire.values())
    elif mode == "sparse":
        assert all(w["form"] == "sparse" for w in wire.values())
    elif scale < 1.0:   # the attractor fills a small part of the image: few segments travel
        assert all(w["form"] == "sparse" and w["fraction_of_dense"] < 0.5 for w in wire.values()), wire
    else:
        assert all(w["form"] == "dense" for w in wire.values()), wire
    for rank, f, c, cs, zs, ss, mx, im in got:
        count[f:f + c], zbuf[f:f + c], steps[f:f + c] = cs, zs, ss
        covered += c
        assert mx == acc.max
        img = im if rank == 0 else img
    assert covered == W * H
    assert np.array_equal(count.reshape(H, W), acc.count)
    assert np.array_equal(zbuf.view(np.uint32).reshape(H, W), acc.zbuf.view(np.uint32))
    assert np.array_equal(steps.view(np.uint64).reshape(H, W), acc.steps.view(np.uint64))
    assert np.array_equal(img, oracle.colorize(cfg, acc))


def test_shard_jobs_partitions_exactly():
    from strange_attractor_renderer_amd.distributed import shard_jobs, slice_of
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
    import strange_attractor_renderer_amd as S
    for npix in (1, 5, 4096, 2048 * 2048, 1800 * 2000, 4096 * 4096):
        for world in (1, 2, 3, 8):
            sp = S.exchange_slice_pixels(npix, world)
            assert sp % 4 == 0 and sp * world >= npix
            assert sum(slice_of(npix, world, r, sp)[1] for r in range(world)) == npix
