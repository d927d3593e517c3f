#!/usr/bin/env python3
"""bench.py — headline benchmark of the render hot path (BASELINE.json): Mrays/s at 4096x4096, 1024 primitives.

  python bench.py [--gpus N --steps K --warmup W]                      one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                        N GPUs, one rank per GPU over RCCL

A step = one frame of the hot path (alt:690-760 of the reference): rebuild the spatial hash from the resident
AABBs, cast one primary and one shadow ray per pixel, shade, quantise — writing the RGBA8 frame and the
palette-index plane. Scene, sprites and output buffers are resident in HBM before the timed region. With N > 1 the
same frame is sharded by row block (cut at bin rows, pixel-art-raytracer_amd/sharding.py) and assembled on rank 0
(strong scaling: total work is fixed as N grows): by default only the tiles that can show a primitive travel (RCCL
point to point) and rank 0 writes the background itself (`--assemble tiles`); `--assemble blocks` gathers whole row
blocks with one RCCL gather per frame; `--assemble none` leaves the frame sharded (render scaling on its own).

Frames in flight: like a swap chain, `--inflight` (default 4) frames are in flight at once, each with its own context,
stream and output buffers (pixel-art-raytracer_amd/pipeline.py); one frame alone is a chain of short latency-bound
kernels that leaves most of the chip idle. `value` is therefore a RATE with 4 frames in flight; the latency of one
frame on its own is reported beside it (`one_frame_at_a_time`).

Timing: W untimed warm-up steps (after a time-based clock ramp), then BLOCKS (30) blocks of exactly K steps, each
bracketed by a barrier + torch.cuda.synchronize() on both sides; `ms_per_step` is the MEDIAN block's time / K (the
spread over the blocks is reported too), with N > 1 the maximum over ranks of each block first.

Mrays/s is nominal = 2 * W * H * frames / seconds (the reference casts exactly one primary and one shadow ray per
pixel, background included: alt:277-279, 703, 738). The GPU path skips the shadow ray of background pixels, whose
colour cannot depend on it (SURVEY a-6); the count actually traced and the rate with every ray traced are reported
beside the headline.

Rank 0 prints ONE JSON line.
"""
import argparse
import datetime
import importlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W = H = L = 4096
N_PRIMS = 1024
SEED = 12345
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BLOCKS = 30
ORACLE_ROWS = (3 * 4096 // 8, 5 * 4096 // 8)  # the band of the frame the CPU baseline renders (and the GPU is checked on)


def cpu_baseline(params, aabbs, light, sprite, rows, gpu_fb, gpu_pal):
    """Oracle (our CPU restatement of the reference, pinned to it) on the host cores: a bounded sample of the same
    workload — the row band `rows` of the same frame, single thread, as the reference runs. The band it renders is
    also what the GPU frame is checked against (`verified`)."""
    from oracle.oracle import Oracle
    o = Oracle()
    grid = o.bin(params, aabbs)
    r0, r1 = rows
    w = params.width
    band = {}

    def run():
        t0 = time.perf_counter()
        gbuf, pal = o.primary(params, grid, sprite, rows=rows)
        fb, _, _ = o.shade(params, grid, gbuf, light, rows=rows)
        band["fb"], band["pal"] = fb, pal
        return time.perf_counter() - t0

    run()  # warm-up, excluded
    times = [run() for _ in range(5)]
    dt = statistics.median(times)
    single = 2.0 * (r1 - r0) * w / dt / 1e6
    verified = bool(np.array_equal(band["fb"][r0 * w:r1 * w].view(np.uint8), gpu_fb[r0 * w * 4:r1 * w * 4]) and
                    np.array_equal(band["pal"][r0 * w:r1 * w], gpu_pal[r0 * w:r1 * w]))
    ncores = os.cpu_count() or 1

    def whole():
        t0 = time.perf_counter()
        o.render(params, aabbs, sprite, light, nthreads=ncores, planes=("fb", "palidx"))
        return time.perf_counter() - t0

    whole()
    dt_all = statistics.median([whole() for _ in range(3)])
    return {
        "value": round(single, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
        "sample": f"rows {r0}..{r1} of the same {params.width}x{params.height} / {len(aabbs)}-primitive frame, single "
                  f"thread, median of 5 after a warm-up ({dt:.2f} s each); faithful to how the reference runs (it has "
                  "no threads)",
        "all_cores": {"value": round(2.0 * params.width * params.height / dt_all / 1e6, 3), "cores": ncores,
                      "sample": f"whole frame, rows split over {ncores} threads, median of 3 ({dt_all:.2f} s each); "
                                "ours, not the reference's"},
    }, verified


def time_blocks(run_block, steps, blocks, barrier, reduce_max):
    """`blocks` blocks of exactly `steps` steps, each bracketed by barrier + synchronize; per block the maximum over
    ranks. Returns the per-step times (ms) of the blocks."""
    out = []
    first = 0
    for _ in range(blocks):
        barrier()
        t0 = time.perf_counter()
        run_block(first, steps)   # ends with torch.cuda.synchronize() (N = 1) / the drained pipeline (N > 1)
        barrier(after_block=True)
        out.append(reduce_max(time.perf_counter() - t0) / steps * 1e3)
        first += steps
    return out


def side_scene(par, pipeline, T, name, params, aabbs, light, sprite, device, depth, steps, counters_key=None):
    """An extra workload beside the headline (same code path, same harness): rate with `depth` frames in flight, one
    frame at a time, and the kernel groups of one frame timed apart."""
    import torch
    w, h = params.width, params.height
    pipe = pipeline.FramePipeline(params, aabbs, sprite, light, depth=depth, device=device, planes=("fb", "palidx"))
    # (a small frame is bound by the host's launches when ONE thread submits for all slots: one thread per slot there,
    # as a C++ host would -- host/par_pipeline.cpp --threads)
    threaded = w * h <= (1 << 21)
    try:
        pipe.submit_many(0, 4 * depth)
        pipe.synchronize()
        ms = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.submit_many(0, steps, threads=threaded)
            pipe.synchronize()
            ms.append((time.perf_counter() - t0) / steps * 1e3)
        per = statistics.median(ms)
        r = pipe.slots[0].renderer
        ptrs = pipe.slots[0].ptrs
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            r.render_device(ptrs, stream=stream)
        torch.cuda.synchronize()
        n1 = max(20, steps // 4)
        t0 = time.perf_counter()
        for _ in range(n1):
            r.render_device(ptrs, stream=stream)
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t0) / n1 * 1e3
        parts = {"hash_build+columns": [], "fill_kernel": [], "render_items_kernel": [], "render_overflow_kernel": []}
        for i in range(13):
            st = r.render_device(ptrs, stream=stream, timed=True)
            if i >= 3:
                parts["hash_build+columns"].append(st.ms_bin)
                parts["fill_kernel"].append(st.ms_fill)
                parts["render_items_kernel"].append(st.ms_render)
                parts["render_overflow_kernel"].append(st.ms_overflow)
        launched = [[] for _ in range(5)]
        for i in range(13):
            st = r.render_device(ptrs, stream=stream, timed=True, flags=par.RENDER_TIMED_AS_LAUNCHED)
            if i >= 3:
                for k in range(5):
                    launched[k].append(st.ms_launch[k])
        launched = [float(np.mean(v)) for v in launched]
        names = launch_names(len(aabbs))
        if st.render_merged:  # small frames: entry items, tile items and overflow columns in one launch
            names[2] = "render_both_kernel"
        dom = max(range(5), key=lambda k: launched[k])  # the longest launch of the frame, as measured here
        clock_ghz = measured_clock_ghz(pipe, depth, row=5 if dom == 3 else 3)
        counters, counters_from = load_counters(counters_key) if counters_key else ({}, None)
        dom_issue = issue_fractions(counters.get(names[dom]), launched[dom], clock_ghz)
        stats = r.stats()
        full = r.render(("palidx",))
        covered = int((full["palidx"] != T.PALIDX_BACKGROUND).sum())
        return {
            "workload": name, "ms_per_frame": round(per, 5), "frames_in_flight": depth,
            "submitting_threads": depth if threaded else 1,
            "mrays_per_s": round(2.0 * w * h / per / 1e3, 1),
            "hbm_frac": round(5.0 * w * h / (per * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "one_frame_at_a_time_ms": round(alone, 5),
            "kernels_ms_one_at_a_time": {k: round(float(np.mean(v)), 5) for k, v in parts.items()},
            "launches_ms_as_launched": dict(zip(names, [round(v, 5) for v in launched])),
            "dominant_launch": dict({"kernel": names[dom], "avg_ms": round(launched[dom], 5), "clock_ghz": clock_ghz,
                                     "counters_from": counters_from,
                                     # the pixel bytes the render kernels have to move, over this launch's time
                                     "hbm_frac": round(5.0 * covered / (launched[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                     if dom in (2, 3) else None}, **dom_issue),
            "covered_pixels": covered, "pixels": w * h, "entities": int(stats.entities),
            "occupied_columns": int(stats.occupied_columns), "overflow_columns": int(stats.overflow_columns),
        }
    finally:
        pipe.close()


def launch_names(n_prims):
    """The launches of a production frame in order (small scenes build the hash in one launch)."""
    build = "build_fill_kernel" if n_prims <= 16384 else "insert_fill_kernel+resolve_fill_kernel"
    return [build, "columns_fill_kernel", "render_items_kernel", "render_tiles_kernel", "render_overflow_kernel"]


def fill_shares():
    """What the library's fill plan gives the hash-build launch and the column launch (par_plan_fill)."""
    pct = int(os.environ.get("PAR_TUNE_FILL_BUILD_PCT", "40"))
    pct = min(100, max(0, pct))
    return pct / 100.0, 1.0 - pct / 100.0


def load_counters(workload):
    """Wave-instruction counts per launch from the round's rocprofv3 passes (tools/profile_round.sh), with the commit
    they were taken at."""
    path = os.path.join(ROOT, "profiles", "sq_counters.json")
    if not os.path.exists(path):
        return {}, None
    with open(path) as f:
        j = json.load(f)
    return j.get(workload, {}), j.get("collected")


def load_traffic():
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return {}, None
    with open(path) as f:
        j = json.load(f)
    return j, j.get("collected")


def issue_fractions(counters, ms, clock_ghz):
    """Share of the chip's vector / scalar issue slots a launch uses: wave-instructions x 4 cycles over 1024 SIMDs."""
    if not counters or not clock_ghz or ms <= 0.0:
        return {}
    cycles = 1024.0 * clock_ghz * 1e9 * ms * 1e-3
    return {"valu_wave_instructions": int(counters.get("SQ_INSTS_VALU", 0)),
            "salu_wave_instructions": int(counters.get("SQ_INSTS_SALU", 0)),
            "valu_frac": round(counters.get("SQ_INSTS_VALU", 0) * 4.0 / cycles, 4),
            "salu_frac": round(counters.get("SQ_INSTS_SALU", 0) * 4.0 / cycles, 4)}


def measured_clock_ghz(pipe, depth, row=3):
    """Shader clock while a render kernel runs (stamp row 3: render_items_kernel, 5: render_tiles_kernel): its
    workgroups' s_memtime spans over their 100 MHz wall-clock spans (debug time stamps of frames rendered with flag
    bit 29; median over workgroups)."""
    import ctypes as C
    pipe.submit_many(0, depth, 1 << 29)
    pipe.synchronize()
    rows, wgs = 6, 8192
    buf = np.zeros(rows * wgs * 8, dtype=np.uint64)
    par = importlib.import_module("pixel-art-raytracer_amd")
    rc = par.lib().par_debug_read_stamps(pipe.slots[0].renderer._ctx, buf.ctypes.data_as(C.c_void_p), buf.size)
    if rc != 0:
        return None
    st = buf.reshape(rows, wgs, 8)[row]
    live = (st[:, 0] > 0) & (st[:, 7] > st[:, 0]) & (st[:, 5] > 0)
    if not live.any():
        return None
    ghz = st[live, 5].astype(np.float64) / ((st[live, 7] - st[live, 0]).astype(np.float64) * 10.0)
    return round(float(np.median(ghz)), 3)


def main():
    global W, H, L, N_PRIMS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--blocks", type=int, default=BLOCKS, help="timed blocks of --steps steps (median reported)")
    ap.add_argument("--inflight", type=int, default=4, help="frames in flight (contexts/streams/buffers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the dense / default-scene side measurements")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for tests)")
    ap.add_argument("--share-gpu", action="store_true", help="tests: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--assemble", choices=("tiles", "blocks", "none"), default="tiles",
                    help="N > 1: how the frame is assembled on rank 0: only the tiles that can show a primitive travel "
                         "and rank 0 writes the background (default), whole row blocks (one RCCL gather), or not at all "
                         "(the frame stays sharded: render scaling on its own)")
    ap.add_argument("--size", type=int, default=W, help="view size (the headline benchmark is 4096; BASELINE config 3: 2048)")
    ap.add_argument("--prims", type=int, default=N_PRIMS, help="primitives (headline 1024; BASELINE config 3: 256)")
    args = ap.parse_args()

    W = H = L = args.size
    N_PRIMS = args.prims
    os.environ.setdefault("PAR_DEBUG_STAMPS", "1")  # allocates the stamp buffers; stamps are taken on request only
    # RCCL between processes needs dmabuf IPC on this host driver (exported by the image; kept if a launcher drops it;
    # it has to be in place before the HIP runtime starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with python -m torch.distributed.run "
                         "--nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            # (the test backend: a rank that is lost fails the job in minutes, not after gloo's default half hour)
            dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=240))

    par = importlib.import_module("pixel-art-raytracer_amd")
    T = par.types
    sharding = importlib.import_module("pixel-art-raytracer_amd.sharding")
    pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")

    params = T.default_params(W, H, L)
    aabbs, light = par.scene_synthetic(N_PRIMS, W, H, L, SEED)
    sprite = par.tile_floor()

    r0, r1 = sharding.row_block(rank, world, H, params.bin_size)
    has_rows = r1 > r0
    gather = None      # "blocks": one FrameGather
    tile_gathers = []  # "tiles": one TileGather per frame slot (each frame in flight has its own inbox and frame)
    rows_alloc = H
    depth = max(1, args.inflight)
    if world > 1:
        rows_alloc = max(sharding.max_block_rows(world, H, params.bin_size), 1)
        if args.assemble == "blocks":
            gather = sharding.FrameGather(H, W * 4, torch.uint8, dev, world, rank, bin_size=params.bin_size)
        elif args.assemble == "tiles":
            # (in place: rank 0 renders its block straight into its rows of the slot's assembled frame)
            tile_gathers = [sharding.TileGather(params, aabbs, dev, world, rank, in_place=True) for _ in range(depth)]
    in_place = bool(tile_gathers) and rank == 0 and has_rows
    pipe = pipeline.FramePipeline(params, aabbs, sprite, light, depth=depth, device=local_rank,
                                  rows=(r0, r1) if has_rows else (0, 1), planes=("fb", "palidx"), rows_alloc=rows_alloc,
                                  fb_targets=[tg.root_block() for tg in tile_gathers] if in_place else None)
    r = pipe.slots[0].renderer
    fb = [s_.buffers["fb"] for s_ in pipe.slots]
    pal = [s_.buffers["palidx"] for s_ in pipe.slots]

    packed = [tg.packed_buffer() for tg in tile_gathers]

    def exchange(slot, k):
        """Behind the render of slot k's block, on the slot's stream: the frame's exchange step."""
        if tile_gathers:
            tg = tile_gathers[k]
            if has_rows:
                tg.pack(slot.buffers["fb"], packed[k], stream=slot.stream.cuda_stream)
            work = tg.exchange(packed[k], async_op=True)
            if rank == 0:
                work.wait()  # (RCCL: the slot's stream waits, not the host)
                tg.assemble(stream=slot.stream.cuda_stream)
            return work
        return gather.gather(slot.buffers["fb"], async_op=True)

    def step(i, flags=0):  # N > 1: render + exchange per frame (the interpreter is not what bounds that path)
        slot = pipe.slot(i)
        with torch.cuda.stream(slot.stream):  # the exchange is ordered after the render on the slot's stream
            if slot.pending is not None:      # the exchange that last read this slot's buffers
                slot.pending.wait()
                slot.pending = None
            if has_rows:
                pipe.submit(i, flags)
            slot.pending = exchange(slot, i % depth)

    def drain():
        for slot in pipe.slots:
            with torch.cuda.stream(slot.stream):
                if slot.pending is not None:
                    slot.pending.wait()
                    slot.pending = None
        pipe.synchronize()
        if gather is not None:
            gather.unpack()
        torch.cuda.synchronize()

    def run_block(first, n, flags=0):
        """Exactly n steps, drained."""
        if gather is None and not tile_gathers:
            # (N = 1, or a frame that stays sharded: nothing but the render)
            pipe.submit_many(first, n, flags)  # the swap chain's loop runs in the library, one call
            torch.cuda.synchronize()           # (one device-wide wait: every slot's stream is done)
        else:
            for i in range(first, first + n):
                step(i, flags)
            drain()

    def barrier(after_block=False):
        if world > 1:
            dist.barrier()
        elif after_block:
            return  # (N = 1: run_block has just synchronised the device; nothing else could be running)
        torch.cuda.synchronize()

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # correctness of what is about to be timed: the frames in flight against one blocking whole-frame render
    run_block(0, depth)
    verified_gpu = None
    hit_pixels = 0
    full = None
    if world > 1 and args.assemble == "none":
        # the frame stays sharded: every rank holds its block against its own render of the whole frame
        mine = r.render(("fb",))["fb"].view(np.uint8)
        ok = all(bool(np.array_equal(fb[k].cpu().numpy()[:(r1 - r0) * W * 4], mine[r0 * W * 4:r1 * W * 4]))
                 for k in range(depth)) if has_rows else True
        verified_gpu = bool(reduce_max(0.0 if ok else 1.0) == 0.0)
    if rank == 0:
        full = r.render(("fb", "palidx"))
        if world > 1 and tile_gathers:
            verified_gpu = all(bool(np.array_equal(tg.frame.cpu().numpy(), full["fb"].view(np.uint8)))
                               for tg in tile_gathers)
        elif world > 1 and gather is not None:
            verified_gpu = bool(np.array_equal(gather.frame.cpu().numpy(), full["fb"].view(np.uint8)))
        elif world > 1:
            pass  # (verified above, on every rank)
        else:
            verified_gpu = all(bool(np.array_equal(fb[k].cpu().numpy(), full["fb"].view(np.uint8)) and
                                    np.array_equal(pal[k].cpu().numpy(), full["palidx"])) for k in range(depth))
        hit_pixels = int((full["palidx"] != T.PALIDX_BACKGROUND).sum())
    barrier()

    # clock ramp (time-based, untimed): the GPU's clocks and the host's caches settle before anything is counted.
    # With N > 1 every iteration holds collectives, so the ranks must run the SAME number of iterations: each rank
    # says whether its own clock wants another one and all of them take the maximum (one small all-reduce per
    # iteration) -- a rank deciding from its own clock alone could run one iteration more than the others and leave
    # the job with unmatched gathers. PAR_BENCH_RAMP_SKEW (tests): rank r asks for that many seconds x r more, which
    # makes the ranks' wishes differ on purpose.
    ramp_s = 0.6 + float(os.environ.get("PAR_BENCH_RAMP_SKEW", "0")) * rank
    t_ramp = time.perf_counter()
    ramp_iterations = 0
    while True:
        run_block(0, 4 * depth)
        ramp_iterations += 1
        more = 1.0 if time.perf_counter() - t_ramp < ramp_s else 0.0
        if reduce_max(more) == 0.0:
            break
    # warm-up (untimed, counted)
    run_block(0, max(args.warmup, 1))

    # timed region: BLOCKS blocks of exactly K steps
    per_step = time_blocks(run_block, args.steps, max(1, args.blocks), barrier, reduce_max)
    ms_per_step = statistics.median(per_step)
    # no kernel flagged a failure in any frame of the timed region (the flag is sticky; stats() raises PAR_ERR_DEVICE)
    for s_ in pipe.slots:
        s_.renderer.stats()

    # the frame's kernels while the timed configuration runs: workgroup time stamps of `depth` frames in mid-flight
    kernels_pipelined = None
    if world == 1:
        pipe.submit_many(0, 6 * depth)
        pipe.submit_many(6 * depth, depth, 1 << 29)
        pipe.submit_many(7 * depth, 6 * depth)
        drain()
        kernels_pipelined = pipe.kernel_spans_us()

    out = None
    if rank == 0:
        rays_per_frame = 2.0 * W * H
        value = rays_per_frame / (ms_per_step * 1e-3) / 1e6
        srt = sorted(per_step)
        out = {
            "metric": f"Mrays/sec at {W}x{H}, {N_PRIMS} prims", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32+f32",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H}x{L} view, bin 40, {N_PRIMS} primitives (splitmix64 seed {SEED}), "
                                   f"light ({5 * W // 8},{H // 2},{L // 4}); RGBA8 frame + palette-index plane",
                       "sharding": f"row blocks over {world} GPU(s), cut at bin rows" +
                                   ("" if world == 1 else
                                    {"tiles": ", the tiles that can show a primitive sent to rank 0 over RCCL (point to "
                                              "point), rank 0 writes the background",
                                     "blocks": ", one RCCL gather of the row blocks to rank 0",
                                     "none": ", the frame left sharded (no exchange)"}[args.assemble]),
                       "frames_in_flight": depth, "streams_overlap_pairwise": bool(pipe.streams_overlap),
                       "note": "value is a rate with frames_in_flight frames in flight; one_frame_at_a_time is the "
                               "latency of a frame on its own"},
            "ms_per_step_spread": {"blocks": len(per_step), "min": round(srt[0], 5), "p10": round(srt[len(srt) // 10], 5),
                                   "median": round(ms_per_step, 5), "p90": round(srt[(9 * len(srt)) // 10], 5),
                                   "max": round(srt[-1], 5), "mean": round(statistics.fmean(per_step), 5)},
            "frames_per_s": round(1e3 / ms_per_step, 1),
            "mpix_per_s": round(W * H / ms_per_step / 1e3, 1),
            "rays": {"nominal_per_frame": int(rays_per_frame), "traced_per_frame": int(W * H + hit_pixels),
                     "note": "shadow rays of background pixels are output-neutral and skipped"},
            "verified_vs_single_gpu_frame": verified_gpu,
        }
        if kernels_pipelined:
            out["kernels_ms_pipelined"] = {k: round(v / 1e3, 5) for k, v in kernels_pipelined.items()}
            out["kernels_ms_pipelined"]["note"] = ("first workgroup start to last workgroup end of each launch of one "
                                                   "frame, mean over the frames in flight, from GPU time stamps taken "
                                                   "while the timed configuration runs (no profiler)")

    # ---- N > 1: where a frame's time goes on this rank, and who took part -----------------------------------
    if world > 1:
        slot = pipe.slots[0]
        reps = 20
        drain()
        t0 = time.perf_counter()
        for i in range(reps):
            if has_rows:
                pipe.submit(0)
            slot.stream.synchronize()
        render_ms = (time.perf_counter() - t0) / reps * 1e3
        barrier()
        gather_ms = 0.0
        if args.assemble != "none":
            t0 = time.perf_counter()
            for i in range(reps):
                with torch.cuda.stream(slot.stream):
                    w_ = exchange(slot, 0)
                    w_.wait()
                slot.stream.synchronize()
            gather_ms = (time.perf_counter() - t0) / reps * 1e3
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: {torch.cuda.get_device_name(local_rank)} (cuda:{local_rank}), "
                                      f"rows {r0}..{r1}")
        ramp_counts = [None] * world
        dist.all_gather_object(ramp_counts, ramp_iterations)
        worst_render = reduce_max(render_ms)
        if rank == 0:
            out["multi_gpu"] = {
                "assemble": args.assemble,
                "render_ms": round(worst_render, 5), "gather_ms": round(gather_ms, 5),
                "gather_bytes_per_rank": (int(rows_alloc * W * 4) if args.assemble == "blocks" else
                                          (max(tile_gathers[0].bytes_sent(q) for q in range(world)) if tile_gathers else 0)),
                "tiles": ({"total": int(len(tile_gathers[0].tiles)), "of": int(np.prod(params.grid_dims()[:2])),
                           "per_rank": [int(c) for c in tile_gathers[0].counts],
                           "bytes_to_rank0": int(sum(tile_gathers[0].bytes_sent(q) for q in range(world)))}
                          if tile_gathers else None),
                "ranks_seen": names,
                "ramp_iterations": ramp_counts,  # (untimed clock ramp: the same on every rank by construction)
                "note": "render_ms: one frame's row block rendered and waited for, one at a time (maximum over ranks); "
                        "gather_ms: one frame's exchange step on its own (pack, send to rank 0, background and unpack "
                        "there; or the gather of the blocks), waited for; the timed region "
                        "overlaps both over the frames in flight",
            }

    # ---- roofline of the dominant kernel + the all-rays-traced rate (N = 1 only; untimed extras) -------------
    if world == 1:
        ptrs = {"fb": fb[0].data_ptr(), "palidx": pal[0].data_ptr()}
        stream = torch.cuda.current_stream().cuda_stream
        # one frame at a time (no other frame in flight): the latency of a frame and its kernels
        for i in range(50):
            r.render_device(ptrs, stream=stream)
        torch.cuda.synchronize()
        lat = []
        for _ in range(5):
            t0 = time.perf_counter()
            for i in range(200):
                r.render_device(ptrs, stream=stream)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) / 200 * 1e3)
        out["one_frame_at_a_time"] = {"ms_per_frame": round(statistics.median(lat), 5)}
        ms = {"bin": [], "fill": [], "render": [], "overflow": []}
        for _ in range(5):
            r.render_device(ptrs, stream=stream, timed=True)
        for _ in range(30):
            st = r.render_device(ptrs, stream=stream, timed=True)
            ms["bin"].append(st.ms_bin)
            ms["fill"].append(st.ms_fill)
            ms["render"].append(st.ms_render)
            ms["overflow"].append(st.ms_overflow)
        avg = {k: float(np.mean(v)) for k, v in ms.items()}
        # ... and the launches of a PRODUCTION frame (the fill riding with the first two), one frame at a time
        names = launch_names(N_PRIMS)
        for _ in range(5):
            r.render_device(ptrs, stream=stream, timed=True, flags=par.RENDER_TIMED_AS_LAUNCHED)
        launched = [[] for _ in range(5)]
        for _ in range(30):
            st = r.render_device(ptrs, stream=stream, timed=True, flags=par.RENDER_TIMED_AS_LAUNCHED)
            for i in range(5):
                launched[i].append(st.ms_launch[i])
        launched = [float(np.mean(v)) for v in launched]
        ncols = int(r.stats().occupied_columns)
        gx, gy, gz = params.grid_dims()
        # Algorithmic bytes: 2.5 B per nominal ray = 5 B per pixel (4 B RGBA8 + 1 B palette index; two
        # rays per pixel). The fill writes every pixel of the frame once (5 B x W x H), shared between the first two
        # launches as the library's plan says (40 % / 60 %); render_items_kernel then writes the pixels primitives
        # cover (5 B x covered pixels) - the only bytes that kernel has to move.
        bytes_frame = 2.5 * 2.0 * W * H
        bytes_render = 5.0 * hit_pixels
        fill_share = fill_shares()
        algorithmic = [bytes_frame * fill_share[0], bytes_frame * fill_share[1], bytes_render, 0.0, 0.0]
        clock_ghz = measured_clock_ghz(pipe, depth)
        counters, counters_from = load_counters("headline")
        traffic_json, traffic_from = load_traffic()
        launches = []
        for i, name in enumerate(names):
            if launched[i] <= 0.0:  # (no such launch in this frame)
                continue
            e = {"kernel": name, "avg_ms": round(launched[i], 5), "algorithmic_bytes": int(algorithmic[i]),
                 "hbm_gbps": round(algorithmic[i] / (launched[i] * 1e-3) / 1e9, 1),
                 "hbm_frac": round(algorithmic[i] / (launched[i] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                 "traffic": traffic_json.get(name + "_hbm_bytes_per_launch")}
            e.update(issue_fractions(counters.get(name), launched[i], clock_ghz))
            launches.append(e)
        dom = max(launches, key=lambda e: e["avg_ms"])  # the longest launch of the frame, as measured in this run
        issue = max(dom.get("valu_frac") or 0.0, dom.get("salu_frac") or 0.0)
        serial_ms = avg["bin"] + avg["fill"] + avg["render"] + avg["overflow"]
        out["roofline"] = {
            "kernel": dom["kernel"],
            "bound": "hbm" if dom["hbm_frac"] >= issue else "valu",
            "achieved": dom["hbm_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dom["hbm_frac"],
            "traffic": dom["traffic"], "traffic_from": traffic_from,
            "valu_frac": dom.get("valu_frac"), "salu_frac": dom.get("salu_frac"),
            "counters_from": counters_from, "clock_ghz": clock_ghz,
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_kernel_ms": dom["avg_ms"],
            "launches": launches,
            "timing": "hipEvent pairs on the launch stream around each launch of a production frame (the fill riding "
                      "with the first two launches, PAR_RENDER_TIMED_AS_LAUNCHED), one frame at a time, mean of 30 "
                      "frames; a bracket includes the launch boundary, about 2 us more than the dispatch duration "
                      "rocprofv3 reports for the same kernel (profiles/*_one_frame_at_a_time_*)",
            "per_unit": "2.5 B per nominal ray = 5 B per pixel (RGBA8 + palette index): the two fill-carrying launches "
                        f"write all {W * H} pixels between them ({fill_share[0]:.0%} / {fill_share[1]:.0%}), "
                        f"render_items_kernel the {hit_pixels} covered ones",
            "issue_roofline": "valu_frac / salu_frac = wave-instructions of the launch (rocprofv3 SQ_INSTS_VALU / "
                              "SQ_INSTS_SALU of the commit named in counters_from) x 4 cycles / (1024 SIMDs x the "
                              "clock measured in this run x the launch's duration): the share of the chip's vector / "
                              "scalar issue slots the launch uses; tools/issuebench.hip has the per-instruction prices "
                              "behind the 4 cycles (profiles/r03_issue_costs.json)",
            "kernels_ms_apart": {"hash_build+columns (bare, no fill riding)": round(avg["bin"], 5),
                                 "fill_kernel (on its own)": round(avg["fill"], 5),
                                 "render_items_kernel": round(avg["render"], 5),
                                 "render_overflow_kernel": round(avg["overflow"], 5)},
            "fill_kernel": {"achieved": round(bytes_frame / (avg["fill"] * 1e-3) / 1e9, 1),
                            "frac": round(bytes_frame / (avg["fill"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "whole_frame": {"achieved": round(bytes_frame / (ms_per_step * 1e-3) / 1e9, 1),
                            "frac": round(bytes_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "ms_per_frame": round(ms_per_step, 5), "kernels_serial_ms": round(serial_ms, 5),
                            # SURVEY 8(d) also counts the whole spatial hash written and read once per frame
                            # (164 B per bin); this design never touches it whole, so `achieved` leaves it out
                            "frac_with_survey_grid_term": round(
                                (bytes_frame + 164.0 * gx * gy * gz + 16.0 * N_PRIMS + 16016.0) /
                                (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "occupied_columns": ncols, "columns": gx * gy,
        }
        # every ray traced, as the reference does (PAR_RENDER_TRACE_BACKGROUND): the shadow ray of a background
        # pixel starts at (x, 0, 0) whatever its row, so it is traced once per x and its result kept per pixel
        # (scratch lit plane); same frames in flight as the headline
        run_block(0, 2 * depth, par.RENDER_TRACE_BACKGROUND)
        k = max(5 * depth, min(1000, args.steps))
        t0 = time.perf_counter()
        run_block(0, k, par.RENDER_TRACE_BACKGROUND)
        dt = time.perf_counter() - t0
        out["rays"]["all_rays_traced_mrays_per_s"] = round(2.0 * W * H * k / dt / 1e6, 1)
        out["rays"]["all_rays_traced_note"] = ("every pixel's shadow ray resolved (lit mask written); the "
                                               f"{W * H - hit_pixels} background rays are {W} distinct rays "
                                               "(one per x), each traced once")
        gpu_fb, gpu_pal = full["fb"].view(np.uint8), full["palidx"]
        pipe.close()
        pipe = None
        if not args.no_extras:
            # the dense regime (every pixel covered, every one of them a traced shadow ray) and the reference's own
            # default scene, through the same harness: extra keys, the headline above is not touched by them
            floor = T.make_aabbs([(i * 20, 0, j * 20, 20, 20, 20) for i in range(W // 20) for j in range(L // 20)])
            out["dense"] = side_scene(par, pipeline, T, f"{W}x{H} full floor of {len(floor)} tiles under the same light: "
                                      "every pixel covered, every shadow ray traced", params, floor, light, sprite,
                                      local_rank, depth, 60, counters_key="floor")
            p0 = T.default_params()
            out["default_scene"] = side_scene(par, pipeline, T, "480x320x320, the reference's graybox world "
                                              "(alt:517-599), light (480,160,80): the reference's own workload",
                                              p0, par.scene_graybox(), T.make_light(480, 160, 80), sprite, local_rank,
                                              depth, 2000, counters_key="graybox")
        if not args.no_cpu_baseline:
            rows = (3 * H // 8, 5 * H // 8)
            out["cpu_baseline"], out["verified_vs_oracle_rows"] = cpu_baseline(params, aabbs, light, sprite, rows,
                                                                               gpu_fb, gpu_pal)
            out["verified_vs_oracle_rows_note"] = (f"rows {rows[0]}..{rows[1]} of the GPU frame (RGBA and palette index) "
                                                   "against the oracle band the cpu_baseline leg renders")

    if rank == 0:
        print(json.dumps(out))
    if pipe is not None:
        pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
