#!/usr/bin/env python3
"""Debug: where does a graph-replayed frame's time go (stage / launch / wait), one frame at a time (GPU box)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
W = 1024
p = T.default_params(W, W, W)
a, l = par.scene_synthetic(512, W, W, W, 77)
fb = torch.zeros(W * W * 4, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
r = par.Renderer(p, 0); r.set_scene(a, par.tile_floor(), l)
r.graph_capture({"fb": fb.data_ptr()}, stream=st.cuda_stream)
rng = np.random.default_rng(5)
vel = rng.choice([-5, 0, 5], size=(512, 3)).astype(np.int16)
def scene(f):
    b = a.copy()
    b["px"] += vel[:, 0] * f; b["py"] += vel[:, 1] * f; b["pz"] += vel[:, 2] * f
    return b
scenes = [scene(f) for f in range(300)]
for label, stage in (("launch only", False), ("stage + launch", True), ("stage moving + launch", 2)):
    ts = [0.0, 0.0, 0.0]
    n = 300
    for f in range(n):
        t0 = time.perf_counter()
        if stage == 2:
            r.graph_stage(scenes[f], 0, l)
        elif stage:
            r.graph_stage(a, 0, l)
        t1 = time.perf_counter()
        r.graph_launch(st.cuda_stream)
        t2 = time.perf_counter()
        st.synchronize()
        t3 = time.perf_counter()
        ts[0] += t1 - t0; ts[1] += t2 - t1; ts[2] += t3 - t2
    print(label, "stage %.1f us  launch %.1f us  wait %.1f us" % tuple(1e6 * x / n for x in ts))
t0 = time.perf_counter()
for f in range(300):
    r.render_device({"fb": fb.data_ptr()}, stream=st.cuda_stream); st.synchronize()
print("direct launch + wait %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
