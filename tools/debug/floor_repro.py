#!/usr/bin/env python3
"""Debug: which pixels / columns differ from the oracle in the floor scene (GPU box)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
par = importlib.import_module("pixel-art-raytracer_amd")
T = par.types
from oracle.oracle import Oracle
o = Oracle()
w, h, l = 800, 600, 600
params = T.default_params(w, h, l)
rows = [(i * 20, 0, j * 20, 20, 20, 20) for i in range(w // 20) for j in range(l // 20)]
rng = np.random.default_rng(4)
rows += [(int(rng.integers(0, w - 20)), int(rng.integers(20, 150)), int(rng.integers(0, l - 20)), 20, 20, 20) for _ in range(300)]
aabbs = T.make_aabbs(rows)
sprite = par.tile_floor()
for lpos in [(500, 300, 150), (40, 20, 580)]:
    light = T.make_light(*lpos)
    exp = o.render(params, aabbs, sprite, light, nthreads=8)
    for planes in (("fb", "palidx"), ("fb", "gbuf", "palidx", "brightness", "lit")):
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            for rep in range(3):
                got = r.render(planes)
                s = r.stats()
                d = (got["fb"].view(np.uint32) != exp["fb"].view(np.uint32)).reshape(h, w)
                cols = sorted(set(zip((np.nonzero(d)[1] // 40).tolist(), (np.nonzero(d)[0] // 40).tolist())))
                print(f"light {lpos} planes {len(planes)} rep {rep}: {int(d.sum())} px differ; occupied {s.occupied_columns} "
                      f"overflow {s.overflow_columns}; differing columns {len(cols)} first {cols[:6]}", flush=True)
