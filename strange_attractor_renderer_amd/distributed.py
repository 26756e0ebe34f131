"""Multi-GPU glue: one process per GPU, trajectories sharded by rank, Runtime::merge (reference
src/lib.rs:708-738) folded in rank order (:1068-1076) over xGMI. PyTorch is plumbing here (device buffers, streams,
torch.distributed == RCCL on ROCm); the arithmetic — the plan of the sparse exchange included — is in libsar_hip.so behind the C
ABI (sar_exchange_*, one context object per runtime and world: api.Exchange).

Two forms of the one exchange step before colorize:

``exchange_colorize`` (sliced, the default of bench.py)
    every rank OWNS one slice of S consecutive pixels. all-to-all of the partial slices (16 B/px: count u32, sortable
    zbuf u32, steps f64 — each pair of GPUs over its own xGMI link), the owner folds the `world` partials of its slice
    with Runtime::merge in rank order, a 4-scalar all-reduce MAX makes max / the depth range global, every rank
    colorizes its slice and only RGBA16 (8 B/px) is gathered on the root. Per rank on the wire:
    16 B * npix * (world-1)/world out, the same in, + 8 B * npix / world to the root.

    SPARSE by default: a frame touches a fifth of its pixels, so only the 64-pixel granules that differ from the reset state
    travel, as 1 KiB records — the ranks all-gather their granule flags (one byte per granule), every rank's library plans
    from them (two block scans on the device, the same on every rank) which records it sends to whom and where the records it
    receives sit, and the all-to-all carries split sizes (the only numbers that come to the host: world x 2 record
    counts, behind ONE wait per frame). A frame whose flags cover more than half the image goes the dense way.

``exchange_merge`` (rooted, two collectives)
    all-reduce(MAX, int64 keys: sortable(z) << 32 | ~rank) + reduce(SUM, int32[3*npix]: count and the two halves of the
    winner's steps bits): rank `dst` ends up holding the complete merged Runtime (20 B/px through ring collectives) — for
    callers that want the merged buffers themselves, not only the image.

Backends: "nccl" (RCCL) moves device buffers directly. With "gloo" (several ranks sharing one GPU, or CPU-only CI) the
same buffers are staged through host memory — the kernels and the protocol are the ones that run under RCCL.
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


def slice_of(npix: int, world: int, rank: int, slice_pixels: int) -> tuple[int, int]:
    """The pixel range [first, first+count) rank `rank` owns in the sliced exchange."""
    first = min(npix, rank * slice_pixels)
    return first, min(npix, first + slice_pixels) - first


def _device_native(dist) -> bool:
    return dist.get_backend() == "nccl"


def _sync_runtime_stream(rt):
    rt.synchronize()


def _all_reduce(dist, t, op, rt):
    if _device_native(dist):
        dist.all_reduce(t, op=op)
    else:  # gloo: through host memory (ranks sharing a GPU, or no GPU at all)
        _sync_runtime_stream(rt)
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)


def _require_shared_stream(dist, rt):
    """The device-native collectives order themselves against torch's CURRENT stream, the pack / merge kernels run on the
    runtime's: unless the two are one stream the collective reads a buffer the pack kernel has not written yet. Loud, not racy."""
    if not _device_native(dist):
        return
    import torch
    if int(rt.stream() or 0) != int(torch.cuda.current_stream().cuda_stream or 0):
        raise RuntimeError("exchange over a device-native backend: the runtime enqueues on another stream than torch's current one; "
                           "call rt.set_stream(torch.cuda.current_stream().cuda_stream) (inside `with torch.cuda.stream(s):`) first")


def _reduce(dist, t, dst, op, rt):
    if _device_native(dist):
        dist.reduce(t, dst=dst, op=op)
    else:
        _sync_runtime_stream(rt)
        h = t.cpu()
        dist.reduce(h, dst=dst, op=op)
        if dist.get_rank() == dst:
            t.copy_(h)


def _all_to_all(dist, out, inp, rt):
    import torch
    if _device_native(dist):
        dist.all_to_all_single(out, inp)
        return
    _sync_runtime_stream(rt)
    world = dist.get_world_size()
    hin = inp.cpu()
    parts = [torch.empty_like(c) for c in hin.chunk(world)]
    # gloo has no all_to_all: `world` scatters (rank r scatters its blocks) — same data movement, test-path only
    for r in range(world):
        dist.scatter(parts[r], list(hin.chunk(world)) if dist.get_rank() == r else None, src=r)
    out.copy_(torch.cat(parts))


def _all_gather(dist, out, t, rt):
    if _device_native(dist):
        dist.all_gather_into_tensor(out, t)
        return
    import torch
    _sync_runtime_stream(rt)
    h = t.cpu()
    parts = [torch.empty_like(h) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, h)
    out.copy_(torch.cat(parts))


def _all_to_all_v(dist, out, inp, out_bytes, in_bytes, rt):
    """all-to-all with split sizes (bytes per peer, in rank order): out / inp are flat uint8 buffers at least as long as the sums."""
    import torch
    if _device_native(dist):
        # (every rank calls the collective, also one with nothing to send or receive: the others may have)
        dist.all_to_all_single(out[: sum(out_bytes)], inp[: sum(in_bytes)], list(out_bytes), list(in_bytes))
        return
    # gloo (ranks sharing a GPU, CPU-only CI): point-to-point through host memory — same data movement, test-path only
    _sync_runtime_stream(rt)
    world, me = dist.get_world_size(), dist.get_rank()
    hin = inp[: sum(in_bytes)].cpu()
    send = list(torch.split(hin, list(in_bytes)))
    recv = [torch.empty(int(n), dtype=torch.uint8) for n in out_bytes]
    reqs = []
    for peer in range(world):
        if peer == me:
            recv[me].copy_(send[me])
            continue
        if out_bytes[peer]:
            reqs.append(dist.irecv(recv[peer], src=peer))
        if in_bytes[peer]:
            reqs.append(dist.isend(send[peer].contiguous(), dst=peer))
    for q in reqs:
        q.wait()
    if sum(out_bytes):
        out[: sum(out_bytes)].copy_(torch.cat(recv))


def _gather(dist, gathered, t, dst, rt):
    import torch
    if _device_native(dist):
        dist.gather(t, list(gathered.chunk(dist.get_world_size())) if dist.get_rank() == dst else None, dst=dst)
        return
    _sync_runtime_stream(rt)
    h = t.cpu()
    if dist.get_rank() == dst:
        parts = [torch.empty_like(h) for _ in range(dist.get_world_size())]
        dist.gather(h, parts, dst=dst)
        gathered.copy_(torch.cat(parts))
    else:
        dist.gather(h, None, dst=dst)


def exchange_merge(ex, dist, key_buf, sum_buf, dst: int = 0):
    """Rooted form: folds every rank's Runtime into rank `dst`'s (rank order == merge order). ex: the rank's api.Exchange; key_buf:
    int64[npix], sum_buf: int32[3*npix] torch tensors on the runtime's device. The pack/unpack kernels run on the RUNTIME's stream
    (a non-blocking stream of its own unless set), the collectives on torch's current stream: give the runtime that
    stream first — ``rt.set_stream(torch.cuda.current_stream().cuda_stream)`` — as bench.py does."""
    rt = ex.runtime
    _require_shared_stream(dist, rt)
    ex.rooted(0, key_buf.data_ptr())
    _all_reduce(dist, key_buf, dist.ReduceOp.MAX, rt)
    ex.rooted(1, key_buf.data_ptr(), sum_buf.data_ptr())
    _reduce(dist, sum_buf, dst, dist.ReduceOp.SUM, rt)
    if ex.rank == dst:
        ex.rooted(2, key_buf.data_ptr(), sum_buf.data_ptr())


class SlicedExchange:
    """Buffers of the sliced exchange for one runtime: allocate once, use every frame."""

    SEG = 64                      # pixels per granule of the sparse form (include/sar.h: SAR_EXCHANGE_GRANULE)
    RECORD = SEG * 16             # bytes per record: count u32 | sortable zbuf u32 | steps f64

    def __init__(self, S_mod, cfg, rt, rank: int, world: int, device, sparse: bool = True, dense_above: float = 0.5):
        import torch
        self.S, self.cfg, self.rt, self.rank, self.world = S_mod, cfg, rt, rank, world
        self.ex = S_mod.Exchange(rt, world, rank)     # the library's context: geometry, plan, pack, fold
        self.sparse = sparse
        self.dense_above = dense_above   # share of touched segments (over all ranks) above which a frame goes the dense way
        self.last = {"form": None, "records_sent": None}
        w, h = rt.dims()
        self.npix = w * h
        self.slice_pixels = self.ex.slice_pixels
        blk = self.slice_pixels * 16
        self.pack = torch.empty(world * blk, dtype=torch.uint8, device=device)
        self.recv = torch.empty(world * blk, dtype=torch.uint8, device=device)
        self.scalars = torch.empty(4, dtype=torch.int64, device=device)
        # RGBA16 travels as bytes (gloo moves no 16-bit integers; RCCL does not care)
        self.rgba_slice = torch.empty(self.slice_pixels * 8, dtype=torch.uint8, device=device)
        # the root's image: `world` slices back to back (a few pixels of padding after npix)
        self.rgba = torch.empty(world * self.slice_pixels * 8, dtype=torch.uint8, device=device)
        self.first, self.count = self.ex.first, self.ex.count
        self.nseg = self.ex.granules
        if self.sparse:
            self.flags = torch.empty(self.nseg, dtype=torch.uint8, device=device)
            self.flags_all = torch.empty(world * self.nseg, dtype=torch.uint8, device=device)

    def bytes_on_the_wire(self) -> dict:
        """Per rank and frame. The sparse form's all-to-all is what the LAST frame sent (its touched segments)."""
        blk = self.slice_pixels * 16
        dense = (self.world - 1) * blk
        out = {"all_to_all_out_per_rank": dense, "gather_to_root_per_rank": self.slice_pixels * 8, "scalars": 32, "form": self.last["form"] or
               ("sparse" if self.sparse else "dense")}
        if self.last["form"] == "sparse":
            out.update({"all_to_all_out_per_rank": self.last["records_sent"] * self.RECORD, "dense_all_to_all_out_per_rank": dense,
                        "granule_flags_all_gather": self.world * self.nseg,
                        "fraction_of_dense": self.last["records_sent"] * self.RECORD / dense if dense else None})
        return out

    def merge(self, dist):
        """Steps 1-3: after this the runtime holds the merged frame inside its own slice and global scalars."""
        rt, ex = self.rt, self.ex
        _require_shared_stream(dist, rt)
        flags_all = None
        if self.sparse:
            ex.flags(self.flags.data_ptr())
            _all_gather(dist, self.flags_all, self.flags, rt)
            flags_all = self.flags_all.data_ptr()
        # the plan (which records go where: two scans on the device), the choice between sparse and dense (every rank holds the same
        # flags and decides alike) and the packing are the library's; the split sizes are the one thing the host waits for
        sparse, send_bytes, recv_bytes = ex.pack(flags_all, self.dense_above, self.pack.data_ptr())
        if sparse:
            _all_to_all_v(dist, self.recv, self.pack, recv_bytes, send_bytes, rt)
            self.last = {"form": "sparse", "records_sent": (sum(send_bytes) - send_bytes[self.rank]) // self.RECORD}
        else:
            _all_to_all(dist, self.recv, self.pack, rt)
            self.last = {"form": "dense", "records_sent": None}
        ex.merge(self.recv.data_ptr(), self.scalars.data_ptr())
        _all_reduce(dist, self.scalars, dist.ReduceOp.MAX, rt)
        ex.finish(self.scalars.data_ptr())

    def colorize(self, dist, dst: int = 0):
        """Step 4: every rank colorizes its slice; the root gathers the image (the first npix*8 bytes of self.rgba there)."""
        import torch
        if self.count:
            self.S.colorize_range_device(self.cfg, self.rt, self.first, self.count, self.rgba_slice.data_ptr())
        _gather(dist, self.rgba, self.rgba_slice, dst, self.rt)
        return self.rgba[: self.npix * 8].view(torch.int16) if self.rank == dst else None


def exchange_colorize(ex: SlicedExchange, dist, dst: int = 0):
    """Sliced form in one call: merge + sharded colorize + gather. Returns the RGBA16 image (flat int16 view of the
    u16 samples, H*W*4) on rank `dst`, None elsewhere."""
    ex.merge(dist)
    return ex.colorize(dist, dst)
