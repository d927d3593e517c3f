#!/usr/bin/env python3
"""Rate of the host-buffer entry point par_render (device frame + PCIe copy back to caller-owned host memory)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
W = H = L = 4096
p = T.default_params(W, H, L)
a, l = par.scene_synthetic(1024, W, H, L, 12345)
r = par.Renderer(p, 0)
r.set_scene(a, par.tile_floor(), l)
for planes in (("fb",), ("fb", "palidx")):
    for _ in range(3):
        r.render(planes)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        r.render(planes)
    dt = (time.perf_counter() - t0) / n
    print(f"par_render {planes}: {dt*1e3:.3f} ms/frame -> {2*W*H/dt/1e6:.0f} Mrays/s (pageable host buffers, PCIe included)")
