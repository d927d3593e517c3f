"""GPU box: random small and mid-size views (every bin size, crowded and sparse, lights inside and outside the view,
dense floors with boxes on them) against the oracle, every plane, for a given number of seconds -- the frames that take
the column teams, the one-launch render of small frames and the tile pass. usage: fuzz_small.py [seconds] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
from oracle.oracle import Oracle
o = Oracle(); sprite = par.tile_floor()
ALL = ("fb", "gbuf", "palidx", "brightness", "lit")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n_cases = 0; n_skipped = 0; n_px = 0
while time.time() - t0 < secs:
    b = int(rng.choice([8, 16, 20, 24, 32, 40, 40, 40, 64, 100, 160]))
    if os.environ.get("FUZZ_BIG"):  # mid-size views: the big frames' launch shapes, tile items of several chunks
        w, h, l = int(rng.integers(64, 256)) * 8, int(rng.integers(400, 2048)), int(rng.integers(200, 2048))
        b = int(rng.choice([20, 32, 40, 40, 40, 64]))
    else:
        w = int(rng.integers(3, 120)) * 8 if rng.random() < 0.7 else int(rng.integers(20, 900))
        h, l = int(rng.integers(30, 700)), int(rng.integers(30, 900))
    kind = int(rng.integers(0, 4))
    if kind == 0:    # sparse random boxes
        n = int(rng.integers(1, 400 if not os.environ.get("FUZZ_BIG") else 3000)); aabbs, light = par.scene_synthetic(n, w, h, l, int(rng.integers(1, 1 << 30)))
    elif kind == 1:  # crowded: boxes folded into a corner
        n = int(rng.integers(50, 600)); aabbs, light = par.scene_synthetic(n, w, h, l, int(rng.integers(1, 1 << 30)))
        aabbs["px"] = (aabbs["px"] % max(3 * b, 60)).astype(aabbs["px"].dtype)
        aabbs["pz"] = (aabbs["pz"] % max(4 * b, 80)).astype(aabbs["pz"].dtype)
    elif kind == 2:  # a floor with boxes of every extent on it
        rows = [(i * 20, 0, j * 20, 20, 20, 20) for i in range((w + 19) // 20) for j in range(max(l // 20, 1))][:6000 if not os.environ.get("FUZZ_BIG") else 12000]
        rows += [(int(rng.integers(0, max(w - 20, 1))), int(rng.integers(0, 150)), int(rng.integers(0, max(l - 20, 1))),
                  int(rng.integers(1, 21)), int(rng.integers(1, 21)), int(rng.integers(1, 21))) for _ in range(int(rng.integers(0, 200)))]
        aabbs = T.make_aabbs(rows); light = T.make_light(int(rng.integers(0, w)), int(rng.integers(20, h)), int(rng.integers(0, l)))
    else:            # a wall along z in one screen column + scattered boxes
        x0 = int(rng.integers(0, max(w - 40, 1)))
        rows = [(x0 + (i % 2) * 10, int(rng.integers(0, 60)), 5 + 30 * i, 20, 20, 20) for i in range(int(rng.integers(5, 40)))]
        rows += [(int(rng.integers(0, max(w - 20, 1))), int(rng.integers(0, 150)), int(rng.integers(0, max(l - 20, 1))), 20, 20, 20) for _ in range(60)]
        aabbs = T.make_aabbs(rows); light = T.make_light(int(rng.integers(-50, w + 50)), int(rng.integers(-50, h + 50)), int(rng.integers(-50, l + 50)))
    if rng.random() < 0.25:
        light = T.make_light(int(rng.integers(-200, w + 200)), int(rng.integers(-200, h + 200)), int(rng.integers(-200, l + 200)))
    try:
        params = T.default_params(w, h, l, b)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            exp = o.render(params, aabbs, sprite, light, nthreads=8)
            for planes in (("fb", "palidx"), ALL):
                out = r.render(planes)
                for k in planes:
                    if out[k].tobytes() != exp[k].tobytes():
                        print("MISMATCH", dict(b=b, w=w, h=h, l=l, kind=kind, n=len(aabbs), plane=k, case=n_cases), flush=True)
                        np.savez(f"gpurun_out/fuzz_fail_{n_cases}.npz", aabbs=aabbs, light=light, w=w, h=h, l=l, b=b)
                        sys.exit(1)
    except par.ParError as e:
        if e.status not in (5, 6):  # unsupported grid / extent: not a parity matter
            raise
        n_skipped += 1
    n_cases += 1
    n_px += w * h
print("fuzz ok:", n_cases, "cases (", n_skipped, "refused as unsupported ),", n_px, "pixels, in", round(time.time() - t0, 1), "s")
