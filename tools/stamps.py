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
if len(sys.argv) > 1 and sys.argv[1] == "floor":
    a = T.make_aabbs([(i * 20, 0, j * 20, 20, 20, 20) for i in range(W // 20) for j in range(L // 20)])
r = par.Renderer(p, 0)
r.set_scene(a, par.tile_floor(), l)
fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
ptrs = {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}
s = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    r.render_device(ptrs, stream=s)
r.render_device(ptrs, stream=s, flags=1 << 29)  # the stamped frame
torch.cuda.synchronize()
n = 6 * 8192 * 8
buf = np.zeros(n, dtype=np.uint64)
L_ = par.lib()
L_.par_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rc = L_.par_debug_read_stamps(r._ctx, buf.ctypes.data_as(C.c_void_p), n)
assert rc == 0, rc
st = buf.reshape(6, 8192, 8).astype(np.float64) * 0.01  # us (100 MHz)
ncol = int(r.stats().occupied_columns)
for row, name in enumerate(["insert+fill", "resolve+fill", "columns+fill", "render_items", "render_overflow"]):
    x = st[row]
    live = x[:, 0] > 0
    if not live.any():
        continue
    b, e = x[live, 0], np.maximum(x[live, 7], x[live, 0])
    t0 = b.min()
    d = e - b
    print(f"{name:16s}: {live.sum():5d} workgroups; starts 0 .. {b.max()-t0:6.2f} us (median {np.median(b)-t0:5.2f}); "
          f"kernel span {e.max()-t0:6.2f} us; workgroup life median {np.median(d):5.2f} p90 {np.percentile(d,90):5.2f} "
          f"max {d.max():5.2f} us")
    if row == 2:
        idx = np.nonzero(live)[0]
        fill = idx >= idx.max() - 255
        print(f"    fill workgroups (last 256): life median {np.median(d[fill]):5.2f} max {d[fill].max():5.2f}; end of the "
              f"last one {e[fill].max()-t0:6.2f}; end of the last column workgroup {e[~fill].max()-t0:6.2f} us")
x3 = st[3]
ok3 = (x3[:, 0] > 0) & (x3[:, 1] > 0) & (x3[:, 2] > 0) & (x3[:, 3] > 0) & (x3[:, 4] > 0) & (x3[:, 7] > 0)
if ok3.any():
    names3 = ["start", "item arrived", "record read", "primary pass done", "shadow test done", "end (stores issued)"]
    idx3 = [0, 1, 2, 3, 4, 7]
    print(f"render wavefront 0 of {ok3.sum()} workgroups that rendered an item:")
    for i in range(1, 6):
        d = x3[ok3, idx3[i]] - x3[ok3, idx3[i - 1]]
        print(f"   {names3[i-1]:>20s} -> {names3[i]:<20s}: median {np.median(d):5.2f}  p90 {np.percentile(d,90):5.2f}  max {d.max():5.2f} us")
    cyc = buf.reshape(6, 8192, 8)[3][ok3, 5].astype(np.float64)
    life = x3[ok3, 7] - x3[ok3, 0]
    print(f"   shader clock while a render wavefront lives: median {np.median(cyc / life):.0f} MHz "
          f"({np.median(cyc):.0f} cycles in {np.median(life):.2f} us)")
x2 = st[2][:ncol]
ok = (x2[:, 5] > 0) & (x2[:, 6] > 0)
print(f"first walk of a column ({ok.sum()} columns): A done -> chain computed median {np.median(x2[ok,5]-x2[ok,2]):.2f} p90 {np.percentile(x2[ok,5]-x2[ok,2],90):.2f}; "
      f"chain -> counts loaded median {np.median(x2[ok,6]-x2[ok,5]):.2f} p90 {np.percentile(x2[ok,6]-x2[ok,5],90):.2f}; "
      f"counts -> walks done median {np.median(x2[ok,3]-x2[ok,6]):.2f} p90 {np.percentile(x2[ok,3]-x2[ok,6],90):.2f} us")
for k, name, labels in ((2, "columns_body", ["start", "listed", "A done", "B walks done", "end"]),):
    x = st[k][:ncol]
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
