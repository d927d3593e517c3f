"""Per-workgroup phase stamps of the headline frame's launches while four frames are in flight (PAR_DEBUG_STAMPS=1):
how long does a wavefront of each kernel live, and in which phase, alone and under load?"""
import ctypes as C, importlib, os, sys
os.environ["PAR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np, torch
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")
W = 4096
p = T.default_params(W, W, W); a, l = par.scene_synthetic(1024, W, W, W, 12345)
for depth in (1, 4):
    pipe = pipeline.FramePipeline(p, a, par.tile_floor(), l, depth=depth)
    pipe.submit_many(0, 40 * depth); pipe.synchronize()
    pipe.submit_many(0, 6 * depth); pipe.submit_many(6 * depth, depth, 1 << 29); pipe.submit_many(7 * depth, 6 * depth)
    pipe.synchronize()
    rows, wgs = 6, 8192
    buf = np.zeros(rows * wgs * 8, dtype=np.uint64)
    print(f"--- {depth} frame(s) in flight")
    for row, name, marks in ((0, "build", (3,)), (2, "columns", (1, 2, 3, 4)), (3, "render", (1, 2, 3, 4))):
        ds, ph, starts, ends = [], {m: [] for m in marks}, [], []
        for s in pipe.slots:
            assert par.lib().par_debug_read_stamps(s.renderer._ctx, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
            st = buf.reshape(rows, wgs, 8)[row].astype(np.int64)
            live = (st[:, 0] > 0) & (st[:, 7] > st[:, 0])
            ds.append((st[live, 7] - st[live, 0]) * 0.01)
            starts.append((st[live, 0] - st[live, 0].min()) * 0.01)
            ends.append((st[live, 7] - st[live, 0].min()) * 0.01)
            for m in marks:
                ok = live & (st[:, m] > 0)
                ph[m].append((st[ok, m] - st[ok, 0]) * 0.01)
        d = np.concatenate(ds)
        so, eo = np.concatenate(starts), np.concatenate(ends)
        print(f"{name:8s} workgroup starts after the launch's first (us): p10 {np.percentile(so, 10):5.1f} p50 {np.percentile(so, 50):5.1f} p90 {np.percentile(so, 90):5.1f} max {so.max():5.1f};  ends: p50 {np.percentile(eo, 50):5.1f} p90 {np.percentile(eo, 90):5.1f} max {eo.max():5.1f}")
        print(f"{name:8s} workgroups {len(d):6d}  lifetime us: median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}   "
              + "  ".join(f"stamp{m}: median {np.median(np.concatenate(ph[m])):5.2f} p90 {np.percentile(np.concatenate(ph[m]), 90):5.2f}" for m in marks if len(np.concatenate(ph[m]))))
    pipe.close()
