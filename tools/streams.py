#!/usr/bin/env python3
"""Which HIP streams overlap? Frames in flight on different subsets of a pool of torch streams (GPU box)."""
import importlib, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
W = H = L = 4096
p = T.default_params(W, H, L)
a, l = par.scene_synthetic(1024, W, H, L, 12345)
NS = 8
streams = [torch.cuda.Stream() for _ in range(NS)]
slots = []
for k in range(4):
    r = par.Renderer(p, 0); r.set_scene(a, par.tile_floor(), l)
    fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    slots.append((r, {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}, fb, pal))

def run(sub, n=800):
    for i in range(40):
        r, ptrs, _, _ = slots[i % len(sub)]
        r.render_device(ptrs, stream=streams[sub[i % len(sub)]].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        r, ptrs, _, _ = slots[i % len(sub)]
        r.render_device(ptrs, stream=streams[sub[i % len(sub)]].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for sub in [(0,), (0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (2, 3), (0, 1, 2), (0, 2, 4), (1, 3, 5), (0, 1, 2, 3), (0, 2, 4, 6), (1, 3, 5, 7), (0, 1, 4, 5), (4, 5, 6, 7)]:
    print(sub, f"{run(sub):6.1f} us/frame", flush=True)
