#!/usr/bin/env python3
"""Is the frame rate host-enqueue bound? One Python thread per pipeline slot (ctypes releases the GIL in HIP calls)."""
import importlib, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
par = importlib.import_module("pixel-art-raytracer_amd")
pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")
T = par.types
W = H = L = 4096
p = T.default_params(W, H, L)
a, l = par.scene_synthetic(1024, W, H, L, 12345)
for depth in (3, 4, 6):
    pipe = pipeline.FramePipeline(p, a, par.tile_floor(), l, depth=depth)
    n = 1500
    for i in range(3 * depth):
        pipe.submit(i)
    pipe.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        pipe.submit(i)
    pipe.synchronize()
    dt = time.perf_counter() - t0
    print(f"depth {depth}: one host thread      {1e6*dt/n:7.1f} us/frame")
    def worker(k, cnt):
        s = pipe.slots[k]
        for _ in range(cnt):
            s.renderer.render_device(s.ptrs, rows=s.rows, stream=s.stream.cuda_stream)
    th = [threading.Thread(target=worker, args=(k, n // depth)) for k in range(depth)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    pipe.synchronize()
    dt = time.perf_counter() - t0
    print(f"depth {depth}: one host thread/slot {1e6*dt/(n//depth*depth):7.1f} us/frame")
    pipe.close()
