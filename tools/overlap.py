#!/usr/bin/env python3
"""Do the bandwidth-bound fill and the latency/VALU-bound pixel work overlap? Stream A renders frames of an EMPTY
scene (hash kernels with nothing to do + the whole 84 MB fill), stream B frames of the headline scene with the fill
switched off (ablation flag bit 28). Alone and together (GPU box)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
W = H = L = 4096
p = T.default_params(W, H, L)
a, l = par.scene_synthetic(1024, W, H, L, 12345)
NO_FILL = 1 << 28
streams = [torch.cuda.Stream() for _ in range(4)]

def slot(aabbs):
    r = par.Renderer(p, 0); r.set_scene(aabbs, par.tile_floor(), l)
    fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    return r, {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}, (fb, pal)

fills = [slot(a[:0]) for _ in range(2)]
rends = [slot(a) for _ in range(2)]

def run(jobs, n=600):
    """jobs: list of (slot, stream index, flags); frame i goes to jobs[i % len(jobs)]."""
    for i in range(40):
        (r, ptrs, _), si, fl = jobs[i % len(jobs)]
        r.render_device(ptrs, stream=streams[si].cuda_stream, flags=fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        (r, ptrs, _), si, fl = jobs[i % len(jobs)]
        r.render_device(ptrs, stream=streams[si].cuda_stream, flags=fl)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

print(f"fill-only frames, one stream             {run([(fills[0], 0, 0)]):6.1f} us per frame")
print(f"fill-only frames, two streams            {run([(fills[0], 0, 0), (fills[1], 1, 0)]):6.1f} us per frame")
print(f"pixel-only frames (no fill), one stream  {run([(rends[0], 2, NO_FILL)]):6.1f} us per frame")
print(f"pixel-only frames, two streams           {run([(rends[0], 2, NO_FILL), (rends[1], 3, NO_FILL)]):6.1f} us per frame")
t = run([(fills[0], 0, 0), (rends[0], 2, NO_FILL)])
print(f"one fill-only + one pixel-only stream    {t:6.1f} us per frame of either kind = {2 * t:6.1f} us per pair")
t = run([(fills[0], 0, 0), (rends[0], 2, NO_FILL), (fills[1], 1, 0), (rends[1], 3, NO_FILL)])
print(f"two fill-only + two pixel-only streams   {t:6.1f} us per frame of either kind = {2 * t:6.1f} us per pair")
