#!/usr/bin/env python3
"""How many columns overflow their record (and render the slow way) in various scenes (GPU box)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types

def run(name, p, a, l):
    with par.Renderer(p, 0) as r:
        r.set_scene(a, par.tile_floor(), l)
        r.render(("fb", "palidx"))
        s = r.stats()
        print(f"{name:40s} entities {s.entities:7d} pairs {s.bin_insertions:7d} columns {s.occupied_columns:6d} overflow {s.overflow_columns:6d}", flush=True)

W = 4096
p = T.default_params(W, W, W)
a, l = par.scene_synthetic(1024, W, W, W, 12345); run("4096 synthetic 1024", p, a, l)
fl = T.make_aabbs([(i * 20, 0, j * 20, 20, 20, 20) for i in range(W // 20) for j in range(W // 20)])
run("4096 full floor", p, fl, T.make_light(2560, 2048, 1024))
run("480x320 graybox", T.default_params(), par.scene_graybox(), T.make_light(480, 160, 80))
for n, w in ((4096, 1024), (16384, 1024), (65536, 2048), (262144, 4096)):
    p = T.default_params(w, w, w)
    a, l = par.scene_synthetic(n, w, w, w, 7)
    run(f"{w} synthetic {n}", p, a, l)
# stacked floors: several layers of tiles (long occluder lists)
w = 1024
p = T.default_params(w, w, w)
st = T.make_aabbs([(i * 20, y, j * 20, 20, 20, 20) for y in (0, 60, 120, 180) for i in range(w // 20) for j in range(w // 20)])
run("1024 four stacked floors", p, st, T.make_light(640, 512, 256))
