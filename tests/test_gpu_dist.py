"""GPU, several ranks on ONE device: the multi-GPU exchange itself — strange_attractor_renderer_amd.distributed
(exchange_merge, the rooted form; SlicedExchange / exchange_colorize, the sliced form) with real processes, real
torch.distributed collectives (gloo: the ranks share cuda:0, so buffers are staged through the host; the kernels and
the protocol are the ones RCCL drives on an 8-GPU node) and the HIP exchange kernels — against Runtime::merge folded in
rank order by the oracle (reference src/lib.rs:708-738, 1068-1076). Also the C-ABI multi-device ParallelRenderer with
one device standing in for several shards."""
import os
import socket
import sys

import numpy as np
import pytest

# SAR_FUZZ_BASE=<k>: the seeded random test of this file draws OTHER cases (tools/soak.sh)
_FUZZ_BASE = 1_000_003 * int(os.environ.get("SAR_FUZZ_BASE", "0"))

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def _worker(rank, world, port, preset, W, H, kind, jobs, n, seed, mode, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import strange_attractor_renderer_amd as S
    from strange_attractor_renderer_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = getattr(S.Config, preset)(iterations=jobs * n, width=W, height=H, jobs_total=jobs, render_kind=kind, scale=1.0,
                                    seed=seed, transparent=0)
    first, cnt = D.shard_jobs(jobs, world, rank)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rt = S.Runtime(cfg, device=0)
        rt.set_stream(stream.cuda_stream)
        S.render_job_range(cfg, rt, n, S.start_points(seed, first, cnt))
        npix = W * H
        if mode == "rooted":
            key = torch.empty(npix, dtype=torch.int64, device="cuda")
            sums = torch.empty(3 * npix, dtype=torch.int32, device="cuda")
            D.exchange_merge(S.Exchange(rt, world, rank), dist, key, sums, dst=0)
            if rank == 0:
                q.put(("rooted", rt.count(), rt.zbuf(), rt.steps(), rt.max(), S.colorize(cfg, rt)))
        else:
            # "sliced": whole slices; "sparse": records of the touched 64-pixel granules whatever their share; "auto": the default
            ex = D.SlicedExchange(S, cfg, rt, rank, world, "cuda", sparse=mode != "sliced", dense_above=2.0 if mode == "sparse" else 0.5)
            img = D.exchange_colorize(ex, dist, dst=0)
            torch.cuda.synchronize()
            # every rank's runtime holds the merged frame inside its own slice
            f, c = ex.first, ex.count
            q.put(("slice", rank, f, c, rt.count().ravel()[f:f + c].copy(), rt.zbuf().ravel()[f:f + c].copy(),
                   rt.steps().ravel()[f:f + c].copy(), rt.max(),
                   img.cpu().numpy().view(np.uint16).reshape(H, W, 4).copy() if rank == 0 else None,
                   ex.bytes_on_the_wire()))
        rt.close()
    dist.barrier()
    dist.destroy_process_group()


def _expected(oracle, preset, W, H, kind, jobs, n, seed, world):
    from strange_attractor_renderer_amd.distributed import shard_jobs
    cfg = getattr(oracle, preset)()
    cfg.width, cfg.height, cfg.scale, cfg.render_kind, cfg.transparent = W, H, 1.0, kind, 0
    parts = []
    for r in range(world):
        first, cnt = shard_jobs(jobs, world, r)
        rt = oracle.Runtime(W, H)
        oracle.render_jobs(cfg, rt, oracle.start_points(seed, first, cnt), n)
        parts.append(rt)
    acc = parts[0]
    for other in parts[1:]:
        assert oracle.merge(acc, other) == 0           # Runtime::merge folded in rank order (:1070-1076)
    return cfg, acc


def _run(world, mode, preset, W, H, kind, jobs, n, seed):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, preset, W, H, kind, jobs, n, seed, mode, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(1 if mode == "rooted" else world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.timeout(600)
@pytest.mark.parametrize("preset,kind", [("poisson_saturne", 0), ("solar_sail", 1)])
def test_rooted_exchange_merge_two_ranks_equals_oracle_merge(sar, oracle, gpu, preset, kind):
    """distributed.exchange_merge itself, 2 ranks on cuda:0: rank 0's merged count / zbuf / steps / max / image equal
    the oracle's per-rank renders folded with Runtime::merge in rank order."""
    W, H, jobs, n, seed, world = 320, 200, 600, 1500, 11, 2
    (_, count, zbuf, steps, mx, img), = _run(world, "rooted", preset, W, H, kind, jobs, n, seed)
    cfg, acc = _expected(oracle, preset, W, H, kind, jobs, n, seed, world)
    assert np.array_equal(count, acc.count) and mx == acc.max
    assert np.array_equal(_bits(zbuf), _bits(acc.zbuf))
    assert np.array_equal(_bits(steps), _bits(acc.steps))
    assert np.array_equal(img, oracle.colorize(cfg, acc))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,preset,kind,form", [(2, "poisson_saturne", 0, "sliced"), (3, "solar_sail", 1, "sliced"), (3, "poisson_saturne", 0, "sparse"),
                                                    (2, "solar_sail", 1, "sparse"), (3, "poisson_saturne", 0, "auto"), (2, "poisson_saturne", 0, "auto")])
def test_sliced_exchange_colorize_equals_oracle_merge(sar, oracle, gpu, world, preset, kind, form):
    """distributed.exchange_colorize, 2 and 3 ranks on cuda:0 (3: uneven last slice, uneven job shards), whole slices or the
    records of the touched granules: the merged slices assembled from the ranks, `max`, and the root's gathered image equal the
    oracle's fold in rank order."""
    W, H, jobs, n, seed = 321, 199, 601, 1200, 17        # odd sizes: npix % world != 0, jobs % world != 0
    got = _run(world, form, preset, W, H, kind, jobs, n, seed)
    cfg, acc = _expected(oracle, preset, W, H, kind, jobs, n, seed, world)
    count = np.zeros(W * H, np.uint32)
    zbuf = np.zeros(W * H, np.float32)
    steps = np.zeros(W * H, np.float64)
    img = None
    covered = 0
    forms = set()
    for _, rank, f, c, cs, zs, ss, mx, im, wire in got:
        count[f:f + c], zbuf[f:f + c], steps[f:f + c] = cs, zs, ss
        covered += c
        assert mx == acc.max                                  # the scalars are global on every rank
        dense = (world - 1) * 16 * sar.exchange_slice_pixels(W * H, world)
        if form == "sliced":
            assert wire["form"] == "dense" and wire["all_to_all_out_per_rank"] == dense
        elif form == "sparse":
            assert wire["form"] == "sparse" and 0 < wire["all_to_all_out_per_rank"] <= dense, wire
        else:   # the default decides by the share of touched granules — the same way on every rank
            forms.add(wire["form"])
            assert wire["all_to_all_out_per_rank"] <= dense and (wire["form"] == "dense" or wire["fraction_of_dense"] <= 0.5 * world / (world - 1) + 1e-9), wire
        if rank == 0:
            img = im
    assert covered == W * H and len(forms) <= 1
    assert np.array_equal(count.reshape(H, W), acc.count)
    assert np.array_equal(_bits(zbuf.reshape(H, W)), _bits(acc.zbuf))
    assert np.array_equal(_bits(steps.reshape(H, W)), _bits(acc.steps))
    assert np.array_equal(img, oracle.colorize(cfg, acc))


@pytest.mark.parametrize("devices,preset,kind", [([0, 0], "poisson_saturne", 0), ([0, 0, 0], "solar_sail", 1)])
def test_c_abi_multi_device_renderer_equals_single_device_and_oracle(sar, oracle, gpu, devices, preset, kind):
    """sar_renderer_new_multi with ONE device listed as several shards (what an 8-GPU node does with 8 devices): job
    slices rendered concurrently on their own streams and host threads, slices exchanged with hipMemcpyPeerAsync,
    folded in device order, colorized per slice straight into the host image. Equal to the single-device renderer and
    to the oracle's fold — for two consecutive frames (the start-point stream runs on, the buffers are reused)."""
    from strange_attractor_renderer_amd.distributed import shard_jobs
    W, H, units, jpu, seed = 333, 211, 96, 5, 23
    cfg = getattr(sar.Config, preset)(iterations=units * jpu * 900 + 77, width=W, height=H, render_kind=kind, scale=1.0,
                                      transparent=0)
    multi = sar.ParallelRenderer(devices=devices, units=units, seed=seed)
    single = sar.ParallelRenderer(device=0, units=units, seed=seed)
    assert multi.num_devices() == len(devices) and multi.num_threads() == units
    dense_bytes = (len(devices) - 1) * 16 * sar.exchange_slice_pixels(W * H, len(devices))
    jobs, n = units * jpu, cfg.iterations // units // jpu
    ocfg = getattr(oracle, preset)()
    ocfg.width, ocfg.height, ocfg.scale, ocfg.render_kind, ocfg.transparent = W, H, 1.0, kind, 0
    for frame in range(3):
        multi.set_exchange([0, 1, 2][frame])          # the default (sparse here: one device has peer access to itself), dense, sparse
        img_m = sar.render_parallel(multi, cfg, jpu)
        img_s = sar.render_parallel(single, cfg, jpu)
        # oracle: the frame's jobs (start points continue across frames), sharded like the devices, folded in order
        parts = []
        for r in range(len(devices)):
            first, cnt = shard_jobs(jobs, len(devices), r)
            rt = oracle.Runtime(W, H)
            oracle.render_jobs(ocfg, rt, oracle.start_points(seed, frame * jobs + first, cnt), n)
            parts.append(rt)
        acc = parts[0]
        for other in parts[1:]:
            oracle.merge(acc, other)
        want = oracle.colorize(ocfg, acc)
        assert np.array_equal(img_m, want)
        assert np.array_equal(img_m, img_s)          # contiguous shards folded in order == the sequential result
        rm = multi.runtime()                          # gathers the merged slices into device 0's runtime
        assert np.array_equal(rm.count(), acc.count) and rm.max() == acc.max
        assert np.array_equal(_bits(rm.zbuf()), _bits(acc.zbuf)) and np.array_equal(_bits(rm.steps()), _bits(acc.steps))
        t = multi.last_timing()
        assert t["n_devices"] == len(devices)
        if frame == 1:
            assert t["exchange_bytes_per_device"] == dense_bytes
        else:   # the records of the touched 64-pixel granules only
            assert 0 < t["exchange_bytes_per_device"] < 0.6 * dense_bytes
    multi.shutdown()
    single.shutdown()


def test_eight_shards_pull_their_slices_on_copy_streams(sar, gpu):
    """The shape of an 8-GPU node on the one GPU there is: 8 shards of device 0 at 2048^2. An owner's 7 pulls run on 7 copy
    streams joined to its stream by events; the frame equals the single-device renderer's bit for bit, into a pageable AND
    into a pinned host image. (One device serves every copy here: what the per-link parallelism buys needs a node with
    several GPUs.)"""
    import torch
    W = H = 2048
    units, jpu = 4096, 4
    cfg = sar.Config.poisson_saturne(iterations=units * jpu * 3000, width=W, height=H, transparent=0)
    single = sar.ParallelRenderer(device=0, units=units, seed=7)
    want = sar.render_parallel(single, cfg, jpu)
    want_count = single.runtime().count().copy()
    single.shutdown()
    ms = {}
    for g in (2, 8):
        multi = sar.ParallelRenderer(devices=[0] * g, units=units, seed=7)
        got = sar.render_parallel(multi, cfg, jpu)
        t = multi.last_timing()
        assert np.array_equal(got, want), f"{g} shards, pageable image"
        assert np.array_equal(multi.runtime().count(), want_count)
        assert t["n_devices"] == g and t["peer_access_failures"] == 0
        dense_bytes = (g - 1) * 16 * sar.exchange_slice_pixels(W * H, g)
        assert 0 < t["exchange_bytes_per_device"] <= 0.25 * dense_bytes    # sparse by default: a fifth of the granules are touched
        multi.set_exchange(1)                                              # the next frame the dense way: whole slices
        sar.render_parallel(multi, cfg, jpu)
        t = multi.last_timing()
        assert t["exchange_bytes_per_device"] == dense_bytes
        ms[g] = t["exchange_ms"]
        multi.shutdown()
    # (exchange_ms counts from a shard's own "packed" event, so with eight shards taking turns on ONE device it mostly
    # measures how long the other seven still render — 0.5 ms with 2 shards, 6.5 ms with 8 here; only a node with several
    # GPUs can say what the links do)
    assert 0.0 < ms[2] < 100.0 and 0.0 < ms[8] < 100.0, ms
    # the same frame into pinned host memory: every device copies its slice straight into the image
    multi = sar.ParallelRenderer(devices=[0] * 8, units=units, seed=7)
    pinned = torch.empty((H, W, 4), dtype=torch.int16).pin_memory()
    sar.render_parallel_into(multi, cfg, jpu, pinned.data_ptr())
    assert np.array_equal(pinned.numpy().view(np.uint16), want)
    multi.shutdown()


def test_multi_device_renderer_on_distinct_gpus(sar, oracle, gpu):
    """The real cross-device path — hipMemcpyPeerAsync between different devices, cross-device stream waits, peer access —
    runs only where the box has more than one GPU (the pool this build ran on hands out one): every GPU once, then twice."""
    ndev = sar.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: the cross-device exchange needs at least two")
    W, H, units, jpu = 1024, 768, 2048, 3
    cfg = sar.Config.poisson_saturne(iterations=units * jpu * 2000, width=W, height=H, transparent=0)
    frames = [cfg.replace(angle=0.4 * f) for f in range(3)]     # from the second on, every device finds its slice announced
    single = sar.ParallelRenderer(device=0, units=units, seed=11)
    want = [sar.render_parallel(single, c, jpu) for c in frames]
    want_count = single.runtime().count().copy()
    single.shutdown()
    assert not np.array_equal(want[0], want[2])
    for devices in (list(range(ndev)), list(range(ndev)) * 2):
        multi = sar.ParallelRenderer(devices=devices, units=units, seed=11)
        for f, (c, w) in enumerate(zip(frames, want)):
            got = sar.render_parallel(multi, c, jpu)
            t = multi.last_timing()
            # the first contact with real peer copies: say what every phase took before judging the pixels
            print(f"[distinct-gpus] devices={devices} frame {f}: " + ", ".join(f"{k}={v:.3f}" if isinstance(v, float) else f"{k}={v}"
                                                                               for k, v in t.items() if not k.startswith("_")), flush=True)
            assert t["peer_access_failures"] == 0, sar.load_library().sar_last_error()
            assert t["n_devices"] == len(devices)
            assert np.array_equal(got, w), (devices, f)
        assert np.array_equal(multi.runtime().count(), want_count), devices
        multi.shutdown()


def _nccl_rank_all_gpus(rank, world, port, W, H, jobs, n, seed, form, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import strange_attractor_renderer_amd as S
    from strange_attractor_renderer_amd import distributed as D
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    cfg = S.Config.poisson_saturne(iterations=jobs * n, width=W, height=H, jobs_total=jobs, seed=seed, transparent=0)
    first, cnt = D.shard_jobs(jobs, world, rank)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rt = S.Runtime(cfg, device=rank)
        rt.set_stream(stream.cuda_stream)
        S.render_job_range(cfg, rt, n, S.start_points(seed, first, cnt))
        if form == "sliced":
            ex = D.SlicedExchange(S, cfg, rt, rank, world, "cuda")
            img = D.exchange_colorize(ex, dist, dst=0)
            torch.cuda.synchronize()
            if rank == 0:
                q.put(("image", img.cpu().numpy().view(np.uint16).reshape(H, W, 4).copy()))
        else:
            key = torch.empty(W * H, dtype=torch.int64, device="cuda")
            sums = torch.empty(3 * W * H, dtype=torch.int32, device="cuda")
            D.exchange_merge(S.Exchange(rt, world, rank), dist, key, sums, dst=0)
            torch.cuda.synchronize()
            if rank == 0:
                q.put(("image", S.colorize(cfg, rt)))
        rt.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("form", ["sliced", "rooted"])
def test_exchange_over_rccl_on_every_gpu_of_the_box(sar, oracle, gpu, form):
    """distributed.py under "nccl" with one rank per GPU of the box, against the oracle's fold in rank order at 512^2 — the
    first run of all_to_all_single / all_reduce / gather between physical GPUs (skips where the box has one)."""
    import multiprocessing as mp
    from strange_attractor_renderer_amd.distributed import shard_jobs
    ndev = sar.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: RCCL between GPUs needs at least two")
    W = H = 512
    jobs, n, seed = 4096 * ndev + 37, 700, 31
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_nccl_rank_all_gpus, args=(r, ndev, port, W, H, jobs, n, seed, form, q)) for r in range(ndev)]
    for p in procs:
        p.start()
    kind, img = q.get(timeout=500)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ocfg = oracle.poisson_saturne()
    ocfg.width, ocfg.height, ocfg.transparent = W, H, 0
    parts = []
    for r in range(ndev):
        first, cnt = shard_jobs(jobs, ndev, r)
        rt = oracle.Runtime(W, H)
        oracle.render_jobs(ocfg, rt, oracle.start_points(seed, first, cnt), n)
        parts.append(rt)
    acc = parts[0]
    for other in parts[1:]:
        oracle.merge(acc, other)
    np.testing.assert_array_equal(img, oracle.colorize(ocfg, acc))


def _nccl_single_rank(port, W, H, jobs, n, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import strange_attractor_renderer_amd as S
    from strange_attractor_renderer_amd import distributed as D
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = S.Config.poisson_saturne(iterations=jobs * n, width=W, height=H, jobs_total=jobs, seed=seed, transparent=0)
    # a runtime left on its own stream: the collective would not wait for its pack kernel — refused, not raced
    stray = S.Runtime(cfg, device=0)
    try:
        D.SlicedExchange(S, cfg, stray, 0, 1, "cuda").merge(dist)
        loud = False
    except RuntimeError as e:
        loud = "another stream" in str(e)
    stray.close()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rt = S.Runtime(cfg, device=0)
        rt.set_stream(stream.cuda_stream)
        S.render_job_range(cfg, rt, n, S.start_points(seed, 0, jobs))
        want = S.colorize(cfg, rt)
        ex = D.SlicedExchange(S, cfg, rt, 0, 1, "cuda")
        img = D.exchange_colorize(ex, dist, dst=0)           # all_to_all_single / all_reduce / gather of RCCL itself
        torch.cuda.synchronize()
        got = img.cpu().numpy().view(np.uint16).reshape(H, W, 4).copy()
        key = torch.empty(W * H, dtype=torch.int64, device="cuda")
        sums = torch.empty(3 * W * H, dtype=torch.int32, device="cuda")
        before = rt.count().copy()
        D.exchange_merge(S.Exchange(rt, 1, 0), dist, key, sums, dst=0)        # rooted form over RCCL
        torch.cuda.synchronize()
        q.put((np.array_equal(got, want), np.array_equal(rt.count(), before), np.array_equal(S.colorize(cfg, rt), want), loud))
        rt.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", range(6))
def test_seeded_random_shard_counts_and_shapes_through_both_exchange_forms(sar, oracle, gpu, seed):
    """2..8 shards of one device, image sizes whose slices end inside a granule and inside a row, job counts that do not divide
    by the shards, both presets and render kinds, view turned and scaled: the frame through the sparse and through the dense
    exchange, bit for bit the sequential oracle's (contiguous shards folded in device order, src/lib.rs:1068-1076)."""
    rng = np.random.default_rng(4000 + seed + _FUZZ_BASE)
    G = int(rng.integers(2, 9))
    preset = ["poisson_saturne", "solar_sail"][int(rng.integers(2))]
    kind = int(rng.integers(2))
    w, h = int(rng.integers(33, 900)), int(rng.integers(17, 700))
    units, jpu, n = int(rng.integers(G, 700)), int(rng.integers(1, 5)), int(rng.integers(40, 700))
    cfg = getattr(sar.Config, preset)(iterations=units * jpu * n + int(rng.integers(units)), width=w, height=h, render_kind=kind,
                                      transparent=int(rng.integers(2)), angle=float(rng.uniform(0, 6.28)),
                                      scale=float(rng.choice([0.5, 1.0, 1.0, 2.2])))
    jobs = units * jpu
    ort = oracle.Runtime(w, h)
    oracle.render_jobs(cfg.replace(jobs_total=jobs).c, ort, sar.start_points(seed + 3, 0, jobs), cfg.iterations // units // jpu)
    want = oracle.colorize(cfg.c, ort)
    for mode in (2, 1):
        pr = sar.ParallelRenderer(devices=[0] * G, units=units, seed=seed + 3)
        pr.set_exchange(mode)
        img = sar.render_parallel(pr, cfg, jpu)
        rm = pr.runtime()
        what = f"seed {seed}: {G} shards, {preset} kind {kind}, {w}x{h}, {units} x {jpu} jobs x {n}, exchange mode {mode}"
        assert np.array_equal(rm.count(), ort.count) and rm.max() == ort.max, what
        assert np.array_equal(_bits(rm.zbuf()), _bits(ort.zbuf)) and np.array_equal(_bits(rm.steps()), _bits(ort.steps)), what
        assert np.array_equal(img, want), what
        pr.shutdown()


@pytest.mark.parametrize("devices,units,jpu", [([0], 1516, 5), ([0, 0, 0], 1516, 5), ([0, 0], 4000, 3)])
def test_frames_announced_by_render_parallel_without_anybody_waiting(sar, oracle, gpu, devices, units, jpu):
    """sar_render_parallel uploads the next frame's start points (two device buffers in turn) and announces them while the
    current frame renders. Frames rendered WITHOUT a host image are not waited for by the call, so the host runs ahead of
    the GPU: the sixth frame (view turned every frame, several launch chunks per shard) must still be the oracle's."""
    n, w, h, frames = 400, 600, 400, 6
    cfg = sar.Config.poisson_saturne(iterations=units * jpu * n, width=w, height=h)
    pr = sar.ParallelRenderer(devices=devices, units=units, seed=9)
    for f in range(frames - 1):
        sar.render_parallel_into(pr, cfg.replace(angle=0.2 * f), jpu, 0)
    c = cfg.replace(angle=0.2 * (frames - 1))
    img = sar.render_parallel(pr, c, jpu)
    ort = oracle.Runtime(w, h)
    oracle.render_jobs(c.replace(jobs_total=units * jpu).c, ort, sar.start_points(9, (frames - 1) * units * jpu, units * jpu), n)
    np.testing.assert_array_equal(img, oracle.colorize(c.c, ort))
    pr.shutdown()


@pytest.mark.timeout(300)
def test_exchange_runs_over_rccl_itself_with_one_rank(sar, gpu):
    """A 1-GPU box cannot hold two RCCL ranks, but it can hold one: the device-native branch of distributed.py
    (all_to_all_single on uint8 blocks, all_reduce MAX on int64, gather, reduce) runs through the real "nccl" backend with
    world size 1 and must leave the frame unchanged. A runtime that enqueues on another stream than torch's current one is refused."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single_rank, args=(_free_port(), 200, 150, 300, 800, 3, q))
    p.start()
    ok = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0 and ok == (True, True, True, True), ok


def test_host_is_off_the_critical_path_of_a_multi_device_frame(sar, oracle, gpu):
    """at 1 048 576 jobs over 8 shards a frame's start points used to be drawn (single-threaded, ~5 ms) and
    uploaded between the render enqueue and the exchange enqueue. Now the next frame's points are drawn on one helper thread
    per device while the GPUs work (the stream is addressable in blocks of 4096 jobs), the four scalars are reduced by the
    devices themselves, and nothing sits between the two enqueues: host_ms_before_exchange stays far below a millisecond —
    and the frames are still the single-device renderer's, bit for bit, announced slices included."""
    W = H = 2048
    units, jpu = 16384, 64                      # 1 048 576 jobs, the job list of an 8-GPU configs[1] frame
    cfg = sar.Config.poisson_saturne(iterations=units * jpu * 120, width=W, height=H, transparent=0)
    frames = [cfg.replace(angle=0.3 * f) for f in range(3)]
    single = sar.ParallelRenderer(device=0, units=units, seed=5)
    want = [sar.render_parallel(single, c, jpu) for c in frames]
    single.shutdown()
    multi = sar.ParallelRenderer(devices=[0] * 8, units=units, seed=5)
    worst = 0.0
    for f, c in enumerate(frames):
        got = sar.render_parallel(multi, c, jpu)
        t = multi.last_timing()
        assert np.array_equal(got, want[f]), f
        worst = max(worst, t["host_ms_before_exchange"])
        assert t["draw_ahead_ms"] > 0.0          # the next frame's points were drawn meanwhile, 131 072 jobs per helper
    assert worst <= 0.5, worst
    # the third frame against the oracle itself (its jobs sit 2 x 1 048 576 jobs into the stream: 512 blocks of 4096)
    ort = oracle.Runtime(W, H)
    oracle.render_jobs_mt(frames[2].replace(jobs_total=units * jpu).c, ort, oracle.start_points(5, 2 * units * jpu, units * jpu), 120)
    np.testing.assert_array_equal(want[2], oracle.colorize(frames[2].c, ort))
    multi.shutdown()


@pytest.mark.parametrize("world,W,H,dense_above", [(2, 512, 384, 2.0), (3, 333, 257, 2.0), (5, 640, 480, 2.0), (8, 1024, 1024, 2.0), (8, 200, 120, 2.0),
                                                   (4, 512, 384, 0.0), (3, 512, 384, 0.5)])
def test_exchange_context_plans_packs_and_folds_like_merge_in_rank_order(sar, oracle, gpu, world, W, H, dense_above):
    """The library's exchange context (sar_exchange_*: flags -> plan -> pack -> merge -> finish) for `world` ranks living in ONE
    process: every rank renders its job slice, the collectives between the steps are done by hand with numpy (all-gather of the
    flags, all-to-all with the split sizes the plan returns, MAX of the scalars) — image sizes whose granule counts are no multiple
    of 64 or of the slice, 2..8 ranks, the sparse form forced (dense_above 2), refused (0) and chosen (0.5). The split sizes must
    agree between every sender and receiver, and every owner's slice must be Runtime::merge folded in rank order (:708-738,
    :1068-1076) by the oracle, bit for bit."""
    import torch
    from strange_attractor_renderer_amd.distributed import shard_jobs
    jobs, n, seed = 97, 900, 12
    cfg = sar.Config.poisson_saturne(iterations=jobs * n, width=W, height=H, jobs_total=jobs, transparent=0)
    rts, orts, exs = [], [], []
    for r in range(world):
        first, cnt = shard_jobs(jobs, world, r)
        st = sar.start_points(seed, first, cnt)
        rt, ort = sar.Runtime(cfg), oracle.Runtime(W, H)
        sar.render_job_range(cfg, rt, n, st)
        oracle.render_jobs(cfg.c, ort, st, n)
        rts.append(rt); orts.append(ort); exs.append(sar.Exchange(rt, world, r))
    ex0 = exs[0]
    nseg, blk = ex0.granules, ex0.block_bytes
    assert nseg == (W * H + 63) // 64 and blk == world * ex0.slice_pixels * 16
    dev = lambda nbytes: torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    flags = [dev(nseg) for _ in range(world)]
    for r in range(world):
        exs[r].flags(flags[r].data_ptr())
        rts[r].synchronize()
    flags_all = torch.cat(flags).contiguous()                                     # the all-gather
    torch.cuda.synchronize()
    send, sb, rb, forms = [], [], [], []
    for r in range(world):
        buf = dev(blk)
        sparse, s_bytes, r_bytes = exs[r].pack(flags_all.data_ptr() if dense_above > 0 else None, dense_above, buf.data_ptr())
        rts[r].synchronize()
        send.append(buf.cpu().numpy()); sb.append(s_bytes); rb.append(r_bytes); forms.append(sparse)
    assert len(set(forms)) == 1                                                   # every rank decides alike
    fa = flags_all.cpu().numpy().reshape(world, nseg)
    touched_share = fa.sum() / (world * nseg)
    assert forms[0] == (dense_above > 0 and touched_share <= dense_above)
    for r in range(world):
        for d in range(world):
            assert sb[r][d] == rb[d][r], (r, d, sb[r][d], rb[d][r])               # what r sends to d is what d expects from r
        if forms[0]:
            sps = ex0.slice_pixels // 64
            own = [int(fa[r, d * sps:(d + 1) * sps].sum()) * 1024 for d in range(world)]
            assert sb[r] == own                                                   # the plan's counts are the flags' (numpy)
    sc = []
    for d in range(world):                                                        # the all-to-all: owner d receives, source by source
        parts = [send[r][sum(sb[r][:d]):sum(sb[r][:d]) + sb[r][d]] for r in range(world)]
        recv = np.zeros(blk, np.uint8)
        cat = np.concatenate(parts) if sum(len(p) for p in parts) else np.zeros(0, np.uint8)
        recv[:len(cat)] = cat
        rbuf = torch.from_numpy(recv).cuda()
        s4 = torch.zeros(4, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        exs[d].merge(rbuf.data_ptr(), s4.data_ptr())
        rts[d].synchronize()
        sc.append(s4.cpu().numpy())
    red = torch.from_numpy(np.max(np.stack(sc), axis=0)).cuda()                   # the all-reduce MAX
    torch.cuda.synchronize()
    acc = orts[0]
    for other in orts[1:]:
        assert oracle.merge(acc, other) == 0
    for d in range(world):
        exs[d].finish(red.data_ptr())
        f, c = exs[d].first, exs[d].count
        assert rts[d].max() == acc.max
        np.testing.assert_array_equal(rts[d].count().ravel()[f:f + c], acc.count.ravel()[f:f + c])
        np.testing.assert_array_equal(_bits(rts[d].zbuf().ravel()[f:f + c]), _bits(acc.zbuf.ravel()[f:f + c]))
        np.testing.assert_array_equal(_bits(rts[d].steps().ravel()[f:f + c]), _bits(acc.steps.ravel()[f:f + c]))
    assert sum(e.count for e in exs) == W * H
    for rt in rts:
        rt.close()                                                                # (closes its exchange context first)


def test_exchange_context_refuses_a_runtime_that_was_resized(sar, gpu):
    """The context's slice geometry and slot tables are those of the image it was made for: after Runtime::set_width_height
    (:667-675) every call answers SAR_ERR_DIM_MISMATCH instead of reading past the buffers; a new context works."""
    import torch
    cfg = sar.Config.poisson_saturne(iterations=1000, width=320, height=200, jobs_total=4, transparent=0)
    rt = sar.Runtime(cfg)
    ex = sar.Exchange(rt, 2, 1)
    flags = torch.zeros(ex.granules, dtype=torch.uint8, device="cuda")
    ex.flags(flags.data_ptr())
    rt.set_width_height(640, 400)
    with pytest.raises(sar.SarError) as err:
        ex.flags(flags.data_ptr())
    assert err.value.status == 2 and "make a new one" in str(err.value)
    ex.close()
    ex = sar.Exchange(rt, 2, 1)
    assert ex.granules == 640 * 400 // 64
    flags = torch.zeros(ex.granules, dtype=torch.uint8, device="cuda")
    ex.flags(flags.data_ptr())
    rt.synchronize()
    ex.close()
    rt.close()
