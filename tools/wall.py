#!/usr/bin/env python3
"""Where does the wall time of a frame go: host enqueue vs device (GPU box)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
W = H = L = 4096
p = T.default_params(W, H, L)
a, l = par.scene_synthetic(1024, W, H, L, 12345)
r = par.Renderer(p, 0)
r.set_scene(a, par.tile_floor(), l)
fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
ptrs = {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}
def run(name, stream_handle, n=300, graph=False):
    for _ in range(20):
        r.graph_launch(stream_handle) if graph else r.render_device(ptrs, stream=stream_handle)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r.graph_launch(stream_handle) if graph else r.render_device(ptrs, stream=stream_handle)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:40s} host enqueue {1e6*(t1-t0)/n:7.1f} us/frame   wall {1e6*(t2-t0)/n:7.1f} us/frame")
run("default (null) stream", torch.cuda.current_stream().cuda_stream)
s1 = torch.cuda.Stream()
run("torch side stream", s1.cuda_stream)
run("default (null) stream again", 0)
r.graph_capture(ptrs, stream=s1.cuda_stream)
run("hipGraph replay on side stream", s1.cuda_stream, graph=True)

# two renderers, two streams, alternating frames (frames in flight = 2)
r2 = par.Renderer(p, 0)
r2.set_scene(a, par.tile_floor(), l)
fb2 = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal2 = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
ptrs2 = {"fb": fb2.data_ptr(), "palidx": pal2.data_ptr()}
s2 = torch.cuda.Stream()
rs = [(r, ptrs, s1), (r2, ptrs2, s2)]
for k in range(2, 5):
    while len(rs) < k:
        rr = par.Renderer(p, 0); rr.set_scene(a, par.tile_floor(), l)
        f_ = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); p_ = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
        rs.append((rr, {"fb": f_.data_ptr(), "palidx": p_.data_ptr()}, torch.cuda.Stream()))
    for i in range(40):
        rr, pp, ss = rs[i % k]; rr.render_device(pp, stream=ss.cuda_stream)
    torch.cuda.synchronize()
    n = 600
    t0 = time.perf_counter()
    for i in range(n):
        rr, pp, ss = rs[i % k]; rr.render_device(pp, stream=ss.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{k} renderers/streams alternating          host enqueue {1e6*(t1-t0)/n:7.1f} us/frame   wall {1e6*(t2-t0)/n:7.1f} us/frame")

# graph replay per slot
for k in (1, 2, 3, 4):
    for (rr, pp, ss) in rs[:k]:
        rr.graph_capture(pp, stream=ss.cuda_stream)
    for i in range(40):
        rr, pp, ss = rs[i % k]; rr.graph_launch(ss.cuda_stream)
    torch.cuda.synchronize()
    n = 600
    t0 = time.perf_counter()
    for i in range(n):
        rr, pp, ss = rs[i % k]; rr.graph_launch(ss.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{k} renderers, hipGraph replay              host enqueue {1e6*(t1-t0)/n:7.1f} us/frame   wall {1e6*(t2-t0)/n:7.1f} us/frame")
