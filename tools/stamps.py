#!/usr/bin/env python3
"""Per-workgroup phase timing of the column records (columns_body) (PAR_DEBUG_STAMPS=1), GPU box."""
import ctypes as C, importlib, os, sys
os.environ["PAR_DEBUG_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
W = H = L = 4096
p = T.default_params(W, H, L)
a, l = par.scene_synthetic(1024, W, H, L, 12345)
r = par.Renderer(p, 0)
r.set_scene(a, par.tile_floor(), l)
fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
ptrs = {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}
s = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    r.render_device(ptrs, stream=s)
torch.cuda.synchronize()
n = 2 * 8192 * 8
buf = np.zeros(n, dtype=np.uint64)
L_ = par.lib()
L_.par_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rc = L_.par_debug_read_stamps(r._ctx, buf.ctypes.data_as(C.c_void_p), n)
assert rc == 0, rc
st = buf.reshape(2, 8192, 8).astype(np.float64) * 0.01  # us (100 MHz)
for k, name, labels in ((0, "columns_body", ["start", "listed", "A done", "B walks done", "end"]),):
    x = st[k]
    live = x[:, 4] > 0
    x = x[live]
    t0 = x[:, 0].min()
    print(f"{name}: {live.sum()} workgroups finished; kernel span {x[:,4].max()-t0:.1f} us")
    print(f"   start times: min 0, median {np.median(x[:,0])-t0:.1f}, max {x[:,0].max()-t0:.1f} us; "
          f"p75 {np.percentile(x[:,0],75)-t0:.1f} p90 {np.percentile(x[:,0],90)-t0:.1f} p99 {np.percentile(x[:,0],99)-t0:.1f}; "
          f"late (>2us): {(x[:,0]-t0>2).sum()}")
    late = np.nonzero(x[:, 0] - t0 > 2)[0]
    ids = np.nonzero(live)[0][late]
    print("   late workgroup ids (first 24):", ids[:24].tolist(), " id%8:", sorted(set((ids % 8).tolist())))
    for i in range(1, 5):
        d = x[:, i] - x[:, i - 1]
        print(f"   {labels[i-1]:>16s} -> {labels[i]:<16s}: median {np.median(d):6.2f}  p90 {np.percentile(d,90):6.2f}  max {d.max():6.2f} us")
    d = x[:, 4] - x[:, 0]
    print(f"   whole workgroup: median {np.median(d):.2f}  p90 {np.percentile(d,90):.2f}  max {d.max():.2f} us")
