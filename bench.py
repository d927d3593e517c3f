#!/usr/bin/env python3
"""bench.py — headline benchmark of the render hot path (BASELINE.json): Mrays/s at 4096x4096, 1024 primitives.

  python bench.py [--gpus N --steps K --warmup W]                      one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                        N GPUs, one rank per GPU over RCCL

A step = one frame of the hot path (alt:690-760 of the reference): rebuild the spatial hash from the resident
AABBs, cast one primary and one shadow ray per pixel, shade, quantise — writing the RGBA8 frame and the
palette-index plane. Scene, sprites and output buffers are resident in HBM before the timed region. With N > 1 the
same frame is sharded by row block (rank r renders rows [H r/N, H (r+1)/N)) and the blocks are gathered to rank 0
with one RCCL gather per frame (strong scaling: total work is fixed as N grows).

Frames in flight: like a swap chain, `--inflight` (default 4) frames are in flight at once, each with its own context,
stream and output buffers (pixel-art-raytracer_amd/pipeline.py); one frame alone is a chain of short latency-bound
kernels that leaves most of the chip idle. `--inflight 1` gives the one-frame-at-a-time rate.

Mrays/s is nominal = 2 * W * H * frames / seconds (the reference casts exactly one primary and one shadow ray per
pixel, background included: alt:277-279, 703, 738). The GPU path skips the shadow ray of background pixels, whose
colour cannot depend on it (SURVEY a-6); the count actually traced and the rate with every ray traced are reported
beside the headline.

Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
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


def cpu_baseline(par, T, params, aabbs, light, sprite, rows):
    """Oracle (our CPU restatement of the reference, pinned to it) on the host cores: a bounded sample of the same
    workload — the row band `rows` of the same 4096x4096 frame, single thread, as the reference runs."""
    from oracle.oracle import Oracle
    import statistics
    o = Oracle()
    grid = o.bin(params, aabbs)
    r0, r1 = rows

    def band():
        t0 = time.perf_counter()
        gbuf, _ = o.primary(params, grid, sprite, rows=rows)
        o.shade(params, grid, gbuf, light, rows=rows)
        return time.perf_counter() - t0

    band()  # warm-up, excluded
    times = [band() for _ in range(5)]
    dt = statistics.median(times)
    single = 2.0 * (r1 - r0) * params.width / dt / 1e6
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
    }


def main():
    global W, H, L, N_PRIMS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--inflight", type=int, default=4, help="frames in flight (contexts/streams/buffers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for tests)")
    ap.add_argument("--share-gpu", action="store_true", help="tests: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--size", type=int, default=W, help="view size (the headline benchmark is 4096; BASELINE config 3: 2048)")
    ap.add_argument("--prims", type=int, default=N_PRIMS, help="primitives (headline 1024; BASELINE config 3: 256)")
    args = ap.parse_args()

    W = H = L = args.size
    N_PRIMS = args.prims
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
            dist.init_process_group(args.backend)

    par = importlib.import_module("pixel-art-raytracer_amd")
    T = par.types
    sharding = importlib.import_module("pixel-art-raytracer_amd.sharding")

    params = T.default_params(W, H, L)
    aabbs, light = par.scene_synthetic(N_PRIMS, W, H, L, SEED)
    sprite = par.tile_floor()
    pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")

    r0, r1 = sharding.row_block(rank, world, H)
    gather = None
    rows_alloc = H
    if world > 1:
        gather = sharding.FrameGather(H, W * 4, torch.uint8, dev, world, rank)
        rows_alloc = gather.max_rows
    depth = max(1, args.inflight)
    pipe = pipeline.FramePipeline(params, aabbs, sprite, light, depth=depth, device=local_rank, rows=(r0, r1),
                                  planes=("fb", "palidx"), rows_alloc=rows_alloc)
    r = pipe.slots[0].renderer
    fb = [s_.buffers["fb"] for s_ in pipe.slots]
    pal = [s_.buffers["palidx"] for s_ in pipe.slots]

    def step(i, flags=0):
        slot = pipe.slot(i)
        if gather is None:
            pipe.submit(i, flags)
            return
        with torch.cuda.stream(slot.stream):  # the collective is ordered after the render on the slot's stream
            if slot.pending is not None:      # the gather that last read this slot's block buffer
                slot.pending.wait()
                slot.pending = None
            pipe.submit(i, flags)
            slot.pending = gather.gather(slot.buffers["fb"], async_op=True)

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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # correctness of what is about to be timed (before the warm-up, so that the GPU does not idle between warm-up
    # and the timed region): rank 0 renders the whole frame alone and compares
    for i in range(depth):
        step(i)
    drain()
    verified = None
    if rank == 0:
        full = r.render(("fb", "palidx"))
        if world > 1:
            verified = bool(np.array_equal(gather.frame.cpu().numpy(), full["fb"].view(np.uint8)))
        else:
            verified = all(bool(np.array_equal(fb[k].cpu().numpy(), full["fb"].view(np.uint8)) and
                                np.array_equal(pal[k].cpu().numpy(), full["palidx"])) for k in range(depth))
        hit_pixels = int((full["palidx"] != T.PALIDX_BACKGROUND).sum())
    barrier()

    # warm-up (untimed)
    for i in range(args.warmup):
        step(i)
    drain()

    # timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays_per_frame = 2.0 * W * H
        value = rays_per_frame * args.steps / elapsed / 1e6
        out = {
            "metric": f"Mrays/sec at {W}x{H}, {N_PRIMS} prims", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32+f32",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H}x{L} view, bin 40, {N_PRIMS} primitives (splitmix64 seed {SEED}), "
                                   f"light ({5 * W // 8},{H // 2},{L // 4}); RGBA8 frame + palette-index plane",
                       "sharding": f"row blocks over {world} GPU(s)" + (", RCCL gather to rank 0" if world > 1 else ""),
                       "frames_in_flight": depth, "streams_overlap_pairwise": bool(pipe.streams_overlap)},
            "frames_per_s": round(args.steps / elapsed, 1),
            "mpix_per_s": round(W * H * args.steps / elapsed / 1e6, 1),
            "rays": {"nominal_per_frame": int(rays_per_frame), "traced_per_frame": int(W * H + hit_pixels),
                     "note": "shadow rays of background pixels are output-neutral and skipped"},
            "verified_vs_single_gpu_frame": verified,
        }

    # ---- roofline of the dominant kernel + the all-rays-traced rate (N = 1 only; untimed extras) -------------
    if world == 1:
        ptrs = {"fb": fb[0].data_ptr(), "palidx": pal[0].data_ptr()}
        stream = torch.cuda.current_stream().cuda_stream
        # one frame at a time (no other frame in flight): the latency of a frame and its kernels
        for i in range(20):
            r.render_device(ptrs, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            r.render_device(ptrs, stream=stream)
        torch.cuda.synchronize()
        out["one_frame_at_a_time"] = {"ms_per_frame": round((time.perf_counter() - t0) / 200 * 1e3, 5)}
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
        ncols = int(r.stats().occupied_columns)
        gx, gy, gz = params.grid_dims()
        # Algorithmic bytes: 2.5 B per nominal ray = 5 B per pixel (4 B RGBA8 + 1 B palette index; two
        # rays per pixel). fill_kernel writes every pixel of the frame once (5 B x W x H); render_wave_kernel then
        # writes the pixels primitives cover (5 B x covered pixels) - the only bytes that kernel has to move.
        bytes_frame = 2.5 * 2.0 * W * H
        bytes_render = 5.0 * hit_pixels
        kernels = {"render_wave_kernel": (avg["render"], bytes_render), "fill_kernel": (avg["fill"], bytes_frame)}
        dominant = max(kernels, key=lambda k: kernels[k][0])
        dom_ms, dom_bytes = kernels[dominant]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(dominant + "_hbm_bytes_per_launch")
        serial_ms = avg["bin"] + avg["fill"] + avg["render"] + avg["overflow"]
        out["roofline"] = {
            "kernel": dominant, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
            "algorithmic_bytes_per_launch": int(dom_bytes), "avg_kernel_ms": round(dom_ms, 5),
            "timing": "hipEvent pairs on the launch stream around each kernel (kernels launched apart and one frame at "
                      "a time for this measurement), mean of 30 frames; the bracket includes the two launch boundaries, "
                      "about 3 us more than the dispatch duration rocprofv3 reports for the same kernel "
                      "(profiles/*_one_frame_at_a_time_*)",
            "per_unit": "2.5 B per nominal ray = 5 B per pixel (RGBA8 + palette index); render_wave_kernel writes "
                        f"the {hit_pixels} covered pixels, fill_kernel all {W * H}",
            "note": "render_wave_kernel is VALU-issue/latency bound, not bandwidth bound (DESIGN.md section 5)",
            "kernels_ms": {"hash_build+columns (3 kernels)": round(avg["bin"], 5),
                           "fill_kernel": round(avg["fill"], 5),
                           "render_wave_kernel": round(avg["render"], 5),
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
        for i in range(2 * depth):
            step(i, par.RENDER_TRACE_BACKGROUND)
        drain()
        k = max(5 * depth, min(1000, args.steps // 2))
        t0 = time.perf_counter()
        for i in range(k):
            step(i, par.RENDER_TRACE_BACKGROUND)
        drain()
        dt = time.perf_counter() - t0
        out["rays"]["all_rays_traced_mrays_per_s"] = round(2.0 * W * H * k / dt / 1e6, 1)
        out["rays"]["all_rays_traced_note"] = ("every pixel's shadow ray resolved (lit mask written); the "
                                               f"{W * H - hit_pixels} background rays are {W} distinct rays "
                                               "(one per x), each traced once")
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(par, T, params, aabbs, light, sprite, (3 * H // 8, 5 * H // 8))

    if rank == 0:
        print(json.dumps(out))
    pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
