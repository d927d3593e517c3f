#!/usr/bin/env python3
"""Timing experiments on the render kernel (GPU box): ablation flags, empty scene, dense floor scene."""
import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types

def run(name, params, aabbs, light, flags=0, planes=("fb", "palidx"), n=30):
    W, H = params.width, params.height
    r = par.Renderer(params, 0)
    r.set_scene(aabbs, par.tile_floor(), light)
    bufs = {"fb": torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"),
            "palidx": torch.zeros(W * H, dtype=torch.uint8, device="cuda"),
            "lit": torch.zeros(W * H, dtype=torch.uint8, device="cuda")}
    ptrs = {k: bufs[k].data_ptr() for k in planes}
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        r.render_device(ptrs, stream=s, flags=flags, timed=True)
    st = [r.render_device(ptrs, stream=s, flags=flags, timed=True) for _ in range(n)]
    ms = [x.ms_render for x in st]; mf = [x.ms_fill for x in st]; mb = [x.ms_bin for x in st]
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(100):
        r.render_device(ptrs, stream=s, flags=flags)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 100
    print(f"{name:44s} render {np.mean(ms)*1e3:8.1f} (min {np.min(ms)*1e3:7.1f})  fill {np.mean(mf)*1e3:6.1f}  bin {np.mean(mb)*1e3:6.1f} us  frame(wall) {wall*1e6:7.1f} us  pairs {r.stats().bin_insertions}")
    r.close()

W = H = L = 4096
p = T.default_params(W, H, L)
a, l = par.scene_synthetic(1024, W, H, L, 12345)
run("4096 synthetic1024", p, a, l)
run("4096 synthetic1024 no-walks(27)", p, a, l, flags=1 << 27)
run("4096 synthetic1024 no-shading(26)", p, a, l, flags=1 << 26)
run("4096 synthetic1024 no-walks no-shading", p, a, l, flags=(1 << 26) | (1 << 27))
run("4096 synthetic1024 no-stores(25)", p, a, l, flags=(1 << 25))
run("4096 synthetic1024 no-primary(24) no-shading", p, a, l, flags=(1 << 24) | (1 << 26))
run("4096 synthetic1024 no-primary no-shading no-stores", p, a, l, flags=(1 << 24) | (1 << 25) | (1 << 26))
run("4096 empty scene (fill only)", p, a[:0], l)
rows = [(i * 20, 0, j * 20, 20, 20, 20) for i in range(W // 20) for j in range(L // 20)]
fl = T.make_aabbs(rows)
run("4096 full floor (41943 prims)", p, fl, l)
run("4096 full floor no-walks", p, fl, l, flags=1 << 27)
p3 = T.default_params()
run("480x320 default graybox", p3, par.scene_graybox(), T.make_light(480, 160, 80))
run("480x320 default graybox no-walks", p3, par.scene_graybox(), T.make_light(480, 160, 80), flags=1 << 27)
