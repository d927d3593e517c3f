#!/usr/bin/env python3
"""Render N frames of one workload (for rocprofv3 --kernel-trace --stats / --pmc runs on the GPU box).
usage: frames.py [synthetic|floor|graybox|small|trace_bg] [frames] [extra render flags, e.g. the ablation bits 24-28]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
what = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
flags = 0
extra = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
if what in ("synthetic", "trace_bg"):
    W = H = L = 4096
    p = T.default_params(W, H, L)
    a, l = par.scene_synthetic(1024, W, H, L, 12345)
    flags = par.RENDER_TRACE_BACKGROUND if what == "trace_bg" else 0
elif what == "floor":
    W = H = L = 4096
    p = T.default_params(W, H, L)
    a = T.make_aabbs([(i * 20, 0, j * 20, 20, 20, 20) for i in range(W // 20) for j in range(L // 20)])
    l = T.make_light(2560, 2048, 1024)
elif what == "small":  # BASELINE config 2
    W = H = L = 512
    p = T.default_params(W, H, L)
    a, l = par.scene_synthetic(64, W, H, L, 12345)
else:
    W, H = 480, 320
    p = T.default_params()
    a, l = par.scene_graybox(), T.make_light(480, 160, 80)
r = par.Renderer(p, 0)
r.set_scene(a, par.tile_floor(), l)
fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
ptrs = {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}
s = torch.cuda.current_stream().cuda_stream
for _ in range(n):
    r.render_device(ptrs, stream=s, flags=flags | extra)
torch.cuda.synchronize()
print("done", what, n, "frames; occupied columns", r.stats().occupied_columns)
