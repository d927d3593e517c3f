import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
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
def t(label, seq):
    t0 = time.perf_counter()
    for s in seq: r.graph_stage(s, 0, l)
    print(label, (time.perf_counter() - t0) / len(seq) * 1e6, "us per stage")
t("static", [a] * 300)
t("same moved scene 5", [scenes[5]] * 300)
t("same moved scene 200", [scenes[200]] * 300)
t("alternating 5/6", [scenes[5], scenes[6]] * 150)
t("sequence", scenes)
t("static again", [a] * 300)
# per-call times of the stage / launch / wait loop: where are the outliers?
ts = []
for f in range(300):
    t0 = time.perf_counter(); r.graph_stage(scenes[f], 0, l)
    t1 = time.perf_counter(); r.graph_launch(st.cuda_stream)
    t2 = time.perf_counter(); st.synchronize()
    t3 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2))
ts = np.array(ts) * 1e6
for k, name in enumerate(("stage", "launch", "wait")):
    v = ts[:, k]
    print(name, "median", np.median(v), "mean", v.mean(), "max", v.max(), "at", int(v.argmax()), "calls over 100 us:", np.nonzero(v > 100)[0][:20].tolist())
