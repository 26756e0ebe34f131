"""Multi-GPU glue: one process per GPU, trajectories sharded by rank, Runtime::merge (reference
src/lib.rs:708-738) folded in rank order (:1068-1076) over xGMI. PyTorch is plumbing here (device buffers, streams,
torch.distributed == RCCL on ROCm); the arithmetic is in libsar_hip.so behind the C ABI (sar_runtime_exchange_*).

Two forms of the one exchange step before colorize:

``exchange_colorize`` (sliced, the default of bench.py)
    every rank OWNS one slice of S consecutive pixels. all-to-all of the partial slices (16 B/px: count u32, sortable
    zbuf u32, steps f64 — each pair of GPUs over its own xGMI link), the owner folds the `world` partials of its slice
    with Runtime::merge in rank order, a 4-scalar all-reduce MAX makes max / the depth range global, every rank
    colorizes its slice and only RGBA16 (8 B/px) is gathered on the root. Per rank on the wire:
    16 B * npix * (world-1)/world out, the same in, + 8 B * npix / world to the root.

    SPARSE by default: a frame touches a fifth of its pixels, so only the 64-pixel granules that differ from the reset state
    travel, as 1 KiB records — the ranks all-gather their granule flags (one byte per granule), every rank derives from
    them (prefix sums where the flags are, the same on every rank) which records it sends to whom and where the records it
    receives sit, and the all-to-all carries split sizes (the only numbers that come to the host: world x 2 record
    counts). A frame whose flags cover more than half the image goes the dense way.

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


def exchange_merge(rt, rank: int, dist, key_buf, sum_buf, dst: int = 0):
    """Rooted form: folds every rank's Runtime into rank `dst`'s (rank order == merge order). key_buf: int64[npix],
    sum_buf: int32[3*npix] torch tensors on the runtime's device. The pack/unpack kernels run on the RUNTIME's stream
    (a non-blocking stream of its own unless set), the collectives on torch's current stream: give the runtime that
    stream first — ``rt.set_stream(torch.cuda.current_stream().cuda_stream)`` — as bench.py does."""
    _require_shared_stream(dist, rt)
    rt.exchange_export(rank, key_buf.data_ptr())
    _all_reduce(dist, key_buf, dist.ReduceOp.MAX, rt)
    rt.exchange_select(rank, key_buf.data_ptr(), sum_buf.data_ptr())
    _reduce(dist, sum_buf, dst, dist.ReduceOp.SUM, rt)
    if rank == dst:
        rt.exchange_import(key_buf.data_ptr(), sum_buf.data_ptr())


class SlicedExchange:
    """Buffers of the sliced exchange for one runtime: allocate once, use every frame."""

    SEG = 64                      # pixels per granule of the sparse form (include/sar.h: SAR_EXCHANGE_GRANULE)
    RECORD = SEG * 16             # bytes per record: count u32 | sortable zbuf u32 | steps f64

    def __init__(self, S_mod, cfg, rt, rank: int, world: int, device, sparse: bool = True, dense_above: float = 0.5):
        import torch
        self.S, self.cfg, self.rt, self.rank, self.world = S_mod, cfg, rt, rank, world
        self.sparse = sparse and hasattr(rt, "exchange_touched")
        self.dense_above = dense_above   # share of touched segments (over all ranks) above which a frame goes the dense way
        self.last = {"form": None, "records_sent": None}
        w, h = rt.dims()
        self.npix = w * h
        self.slice_pixels = S_mod.exchange_slice_pixels(self.npix, world)
        blk = self.slice_pixels * 16
        self.pack = torch.empty(world * blk, dtype=torch.uint8, device=device)
        self.recv = torch.empty(world * blk, dtype=torch.uint8, device=device)
        self.scalars = torch.empty(4, dtype=torch.int64, device=device)
        # RGBA16 travels as bytes (gloo moves no 16-bit integers; RCCL does not care)
        self.rgba_slice = torch.empty(self.slice_pixels * 8, dtype=torch.uint8, device=device)
        # the root's image: `world` slices back to back (a few pixels of padding after npix)
        self.rgba = torch.empty(world * self.slice_pixels * 8, dtype=torch.uint8, device=device)
        self.first, self.count = slice_of(self.npix, world, rank, self.slice_pixels)
        self.nseg = (self.npix + self.SEG - 1) // self.SEG
        self.sps = self.slice_pixels // self.SEG   # granules per slice (the slice is whole granules)
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

    def slot_tables(self, flags_all):
        """From every rank's granule flags ([world][nseg] bool tensor, wherever it lives): where my records go (send_slot[nseg],
        owner by owner, granule by granule), how many to each owner, where the records I receive sit (recv_slot[world * sps],
        source by source) and how many come from each source. The same arithmetic on every rank; prefix sums on the flags' device."""
        import torch
        world, sps, rank = self.world, self.sps, self.rank
        fp = torch.zeros((world, world * sps), dtype=torch.bool, device=flags_all.device)
        fp[:, : self.nseg] = flags_all
        mine = fp[rank]
        minus1 = torch.tensor(-1, dtype=torch.int32, device=fp.device)
        send_slot = torch.where(mine, (torch.cumsum(mine, 0) - 1).to(torch.int32), minus1)[: self.nseg].contiguous()
        sub = fp[:, rank * sps:(rank + 1) * sps].reshape(-1)
        recv_slot = torch.where(sub, (torch.cumsum(sub, 0) - 1).to(torch.int32), minus1).contiguous()
        counts = torch.cat([mine.reshape(world, sps).sum(dim=1), sub.reshape(world, sps).sum(dim=1), flags_all.sum().reshape(1)]).cpu()  # (the one host copy)
        return send_slot, [int(c) for c in counts[:world]], recv_slot, [int(c) for c in counts[world:2 * world]], int(counts[-1])

    def merge(self, dist):
        """Steps 1-3: after this the runtime holds the merged frame inside its own slice and global scalars."""
        import torch
        rt = self.rt
        _require_shared_stream(dist, rt)
        sparse = self.sparse
        if sparse:
            rt.exchange_touched(self.flags.data_ptr())
            _all_gather(dist, self.flags_all, self.flags, rt)
            flags_all = self.flags_all.reshape(self.world, self.nseg) != 0
            send_slot, send_counts, recv_slot, recv_counts, touched = self.slot_tables(flags_all)
            # a frame that covers the image goes the dense way (every rank sees the same flags, so every rank decides alike)
            sparse = touched <= self.dense_above * self.world * self.nseg
        if sparse:
            self._slots = (send_slot, recv_slot)   # (alive until the kernels that read them have run)
            rt.exchange_pack_sparse(send_slot.data_ptr(), self.pack.data_ptr())
            _all_to_all_v(dist, self.recv, self.pack, [c * self.RECORD for c in recv_counts], [c * self.RECORD for c in send_counts], rt)
            rt.exchange_merge_sparse(self.world, self.rank, recv_slot.data_ptr(), self.recv.data_ptr())
            self.last = {"form": "sparse", "records_sent": sum(send_counts) - send_counts[self.rank]}
        else:
            rt.exchange_pack(self.world, self.pack.data_ptr())
            _all_to_all(dist, self.recv, self.pack, rt)
            rt.exchange_merge_slices(self.world, self.rank, self.recv.data_ptr())
            self.last = {"form": "dense", "records_sent": None}
        rt.exchange_scalars_export(self.scalars.data_ptr())
        _all_reduce(dist, self.scalars, dist.ReduceOp.MAX, rt)
        rt.exchange_scalars_import(self.scalars.data_ptr())

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
