#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: 1024x1024, 512 moving primitives, 300 frames (GPU box). Frames per second of
(a) the captured hipGraph (stage + launch, one frame at a time), (b) direct launches, one frame at a time,
(c) four frames in flight (each slot's context gets the frame's AABBs before it renders)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
par = importlib.import_module("pixel-art-raytracer_amd")
pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")
T = par.types
W = H = L = 1024
N, FRAMES = 512, 300
p = T.default_params(W, H, L)
a0, l0 = par.scene_synthetic(N, W, H, L, 77)
rng = np.random.default_rng(5)
vel = rng.choice([-5, 0, 5], size=(N, 3)).astype(np.int16)  # the reference's step size (alt:643-678)

def scene(f):
    a = a0.copy()
    a["px"] += vel[:, 0] * f; a["py"] += vel[:, 1] * f; a["pz"] += vel[:, 2] * f
    return a

scenes = [scene(f) for f in range(FRAMES)]
fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
ptrs = {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}
stream = torch.cuda.Stream()
r = par.Renderer(p, 0); r.set_scene(a0, par.tile_floor(), l0)
r.graph_capture(ptrs, stream=stream.cuda_stream)
for mode in ("graph", "direct"):
    for rep in range(2):
        t0 = time.perf_counter()
        for f in range(FRAMES):
            if mode == "graph":
                r.graph_stage(scenes[f], 0, l0)
                r.graph_launch(stream.cuda_stream)
                stream.synchronize()
            else:
                r.update_aabbs(scenes[f], 0)
                r.render_device(ptrs, stream=stream.cuda_stream)
        stream.synchronize()
        dt = time.perf_counter() - t0
    print(f"{mode:8s} {FRAMES / dt:9.0f} frames/s  {1e6 * dt / FRAMES:7.1f} us/frame  {2.0 * W * H * FRAMES / dt / 1e6:9.0f} Mrays/s")
r.close()
pipe = pipeline.FramePipeline(p, a0, par.tile_floor(), l0, depth=4)
for rep in range(2):
    t0 = time.perf_counter()
    for f in range(FRAMES):
        pipe.update_aabbs(f, scenes[f])
        pipe.submit(f)
    pipe.synchronize()
    dt = time.perf_counter() - t0
print(f"{'4 in flight':8s} {FRAMES / dt:9.0f} frames/s  {1e6 * dt / FRAMES:7.1f} us/frame  {2.0 * W * H * FRAMES / dt / 1e6:9.0f} Mrays/s")
pipe.close()

# (d) four contexts, each with its own captured graph (stage + launch; the slot's previous frame has to be done
# before its staging area is rewritten)
slots = []
for k in range(4):
    rr = par.Renderer(p, 0); rr.set_scene(a0, par.tile_floor(), l0)
    fbk = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); palk = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    rr.graph_capture({"fb": fbk.data_ptr(), "palidx": palk.data_ptr()}, stream=st.cuda_stream)
    slots.append((rr, st, fbk, palk))
for rep in range(2):
    t0 = time.perf_counter()
    for f in range(FRAMES):
        rr, st, _, _ = slots[f % 4]
        st.synchronize()
        rr.graph_stage(scenes[f], 0, l0)
        rr.graph_launch(st.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{'4 graphs in flight':8s} {FRAMES / dt:9.0f} frames/s  {1e6 * dt / FRAMES:7.1f} us/frame  {2.0 * W * H * FRAMES / dt / 1e6:9.0f} Mrays/s")
