"""Which render workgroups of a small frame are the slow ones? Workgroup time stamps (PAR_DEBUG_STAMPS=1) of one
graybox frame rendered alone: histogram of workgroup durations, the slowest ones, and the phase stamps inside."""
import ctypes as C, importlib, os, sys
os.environ["PAR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np, torch
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
W, H = 480, 320
p = T.default_params(); a, l = par.scene_graybox(), T.make_light(480, 160, 80)
r = par.Renderer(p, 0); r.set_scene(a, par.tile_floor(), l)
fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
ptrs = {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}
s = torch.cuda.current_stream().cuda_stream
for _ in range(50): r.render_device(ptrs, stream=s)
torch.cuda.synchronize()
r.render_device(ptrs, stream=s, flags=1 << 29)
torch.cuda.synchronize()
rows, wgs = 6, 8192
buf = np.zeros(rows * wgs * 8, dtype=np.uint64)
assert par.lib().par_debug_read_stamps(r._ctx, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(rows, wgs, 8)
for row, name in ((2, "columns"), (3, "render")):
    b, e = st[row, :, 0].astype(np.int64), st[row, :, 7].astype(np.int64)
    live = b > 0
    t0 = b[live].min()
    d = (e[live] - b[live]) * 0.01
    idx = np.nonzero(live)[0]
    print(name, "workgroups", live.sum(), "span us", (e[live].max() - t0) * 0.01, "durations us: median", np.median(d), "p90", np.percentile(d, 90), "max", d.max())
    print("  histogram (us):", np.histogram(d, bins=[0, 1, 2, 3, 4, 6, 8, 10, 12, 16, 32])[0].tolist())
    order = np.argsort(-d)[:8]
    for o in order:
        w = idx[o]
        print(f"  wg {w}: start {(b[w]-t0)*0.01:.2f} dur {d[o]:.2f}  phase stamps", [(int(st[row, w, k]) - int(b[w])) * 0.01 if st[row, w, k] else None for k in range(1, 5)])
stats = r.stats()
print("occupied columns", stats.occupied_columns, "overflow", stats.overflow_columns)
