import ctypes as C, importlib, os, sys
os.environ["PAR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np, torch
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")
W = 4096
p = T.default_params(W, W, W); a, l = par.scene_synthetic(1024, W, W, W, 12345)
depth = 4
pipe = pipeline.FramePipeline(p, a, par.tile_floor(), l, depth=depth)
pipe.submit_many(0, 40 * depth); pipe.synchronize()
for rep in range(3):
    pipe.submit_many(0, 6 * depth); pipe.submit_many(6 * depth, depth, 1 << 29); pipe.submit_many(7 * depth, 6 * depth)
    pipe.synchronize()
    rows, wgs = 6, 8192
    buf = np.zeros(rows * wgs * 8, dtype=np.uint64)
    for k, s in enumerate(pipe.slots):
        par.lib().par_debug_read_stamps(s.renderer._ctx, buf.ctypes.data_as(C.c_void_p), buf.size)
        st = buf.reshape(rows, wgs, 8)[3].astype(np.int64)
        live = (st[:, 0] > 0) & (st[:, 7] > st[:, 0])
        idx = np.nonzero(live)[0]
        t0 = st[live, 0].min()
        ends = (st[live, 7] - t0) * 0.01; starts = (st[live, 0] - t0) * 0.01
        per = [(ends[idx % 8 == x].max(), np.median(starts[idx % 8 == x])) for x in range(8)]
        print(f"rep {rep} slot {k} render: span {ends.max():5.1f} us; per XCD (last end / median start):", " ".join(f"{e:4.0f}/{s_:4.1f}" for e, s_ in per))
pipe.close()
