"""GPU parity, second batch: hipGraph replay with moving primitives (BASELINE config 5), the paths that leave the
wave kernel (column records that overflow, shadow rays that start in unoccupied bins), mouse pick."""
import os

import numpy as np
import pytest

from test_gpu_parity import ALL, assert_planes_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sprite(par):
    return par.tile_floor()


def test_graph_replay_with_moving_primitives(par, oracle, sprite, T):
    import torch
    w, h, l = 640, 400, 400
    params = T.default_params(w, h, l)
    n = 256
    aabbs, light = par.scene_synthetic(n, w, h, l, 31)
    rng = np.random.default_rng(5)
    vel = rng.choice([-5, 0, 5], size=(n, 3))  # the reference's step size (alt:643-678)
    fb = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    pal = torch.zeros(w * h, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        r.graph_capture({"fb": fb.data_ptr(), "palidx": pal.data_ptr()}, stream=stream.cuda_stream)
        for f in range(12):
            if f:
                aabbs["px"] += vel[:, 0].astype(np.int16)
                aabbs["py"] += vel[:, 1].astype(np.int16)
                aabbs["pz"] += vel[:, 2].astype(np.int16)
                light["x"] -= 5
                r.graph_stage(aabbs, 0, light)
            r.graph_launch(stream.cuda_stream)
            stream.synchronize()
            exp = oracle.render(params, aabbs, sprite, light, planes=("fb", "palidx"))
            assert np.array_equal(fb.cpu().numpy(), exp["fb"].view(np.uint8)), f"frame {f}"
            assert np.array_equal(pal.cpu().numpy(), exp["palidx"]), f"frame {f}"
        # staging keeps the host's footprints and totals exact but lets the per-column histograms lag: a frame
        # rendered the ordinary way right after must come out the same (sized for an overflow list it cannot rule out),
        # and so must one rendered after a blocking update has brought the histograms up to date
        got = r.render(("fb", "palidx"))
        assert np.array_equal(got["fb"].view(np.uint8), exp["fb"].view(np.uint8)), "blocking render after staging"
        assert np.array_equal(got["palidx"], exp["palidx"]), "blocking render after staging"
        aabbs["px"] += vel[:, 0].astype(np.int16)
        r.update_aabbs(aabbs, 0)
        exp = oracle.render(params, aabbs, sprite, light, planes=("fb", "palidx"))
        got = r.render(("fb", "palidx"))
        assert np.array_equal(got["fb"].view(np.uint8), exp["fb"].view(np.uint8)), "after a blocking update"
        assert np.array_equal(got["palidx"], exp["palidx"]), "after a blocking update"


def test_update_aabbs_between_frames(par, oracle, sprite, T):
    w, h, l = 480, 320, 320
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(300, w, h, l, 8)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for f in range(6):
            aabbs["px"][:50] += 5
            aabbs["pz"][10:60] -= 5
            r.update_aabbs(aabbs[:60], 0)
            out = r.render(ALL)
            assert_planes_equal(out, oracle.render(params, aabbs, sprite, light), ALL, f"frame {f}")


def test_overflowing_columns_take_the_overflow_kernel(par, oracle, sprite, T):
    # hundreds of boxes stacked into a few columns: more slot records / occluders than a column record holds
    w, h, l = 480, 320, 640
    params = T.default_params(w, h, l)
    rng = np.random.default_rng(2)
    rows = [(int(rng.integers(200, 260)), int(rng.integers(0, 40)), int(z), 20, 20, 20)
            for z in rng.integers(0, 300, 500)]
    rows += [(i * 20, 0, j * 20, 20, 20, 20) for i in range(24) for j in range(16) if not 4 <= i < 8]
    # seven boxes in each of the sixteen z-bins of ONE screen column (y + z constant keeps them on the same rows):
    # 112 slot records where a column record holds 64
    rows += [(100 + k, 290 - 40 * b, 40 * b + 10, 20, 20, 20) for b in range(16) for k in range(7)]
    aabbs = T.make_aabbs(rows)
    for lpos in [(300, 160, 80), (230, 60, 10)]:
        light = T.make_light(*lpos)
        exp = oracle.render(params, aabbs, sprite, light)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            fast = r.render(("fb", "palidx", "brightness", "gbuf"))
            assert_planes_equal(fast, exp, ("fb", "palidx", "brightness", "gbuf"), f"overflow {lpos}")
            assert r.stats().overflow_columns > 0, "the scene is meant to overflow some column records"
            assert_planes_equal(r.render(ALL), exp, ALL, f"overflow dense {lpos}")


def test_long_shadow_walks_overflow_the_stage(par, oracle, sprite, T):
    # a row of full bins between the primitives and the light: the walk from the far end collects more occluder
    # records (11 bins x 7) than a start bin's list (64) holds, so the pixels starting there trace their shadow
    # rays with trace_hash_for_light as written, per lane (the column keeps its record); PAR_FORCE_GENERIC
    # (test_overflow_kernel_on_every_column) takes the same scene through the in-kernel stage, which overflows too
    w, h, l = 480, 320, 320
    params = T.default_params(w, h, l)
    rows = [(40 * bx + 2 * k, 100, 100, 20, 20, 20) for bx in range(12) for k in range(7)]
    rows += [(i * 20, 0, j * 20, 20, 20, 20) for i in range(24) for j in range(16)]
    aabbs = T.make_aabbs(rows)
    for lpos in [(470, 110, 110), (475, 118, 102), (5, 110, 110)]:
        light = T.make_light(*lpos)
        exp = oracle.render(params, aabbs, sprite, light)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(ALL), exp, ALL, f"long walk {lpos}")
            assert_planes_equal(r.render(("fb", "palidx")), exp, ("fb", "palidx"), f"long walk {lpos} (riding fill)")


def test_shadow_rays_from_unoccupied_bins(par, oracle, T):
    # sprite depths far outside the box (and negative world z) put the ray's start bin where no primitive is:
    # the wave kernel then walks per lane (trace_hash_for_light as written)
    w, h, l = 480, 320, 320
    params = T.default_params(w, h, l)
    sprite = par.tile_floor()
    sprite["depth"][0][:400] = 95      # top face "floats" two bins deeper
    sprite["depth"][0][400:] = -70     # front face two bins nearer (negative world z near the view front)
    aabbs, light = par.scene_synthetic(250, w, h, l, 13)
    aabbs["pz"][:40] = -20
    exp = oracle.render(params, aabbs, sprite, light)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        fast = r.render(("fb", "palidx", "brightness", "gbuf"))
        assert_planes_equal(fast, exp, ("fb", "palidx", "brightness", "gbuf"), "exotic depth")
        assert_planes_equal(r.render(ALL), exp, ALL, "exotic depth dense")


def test_pick_and_stats(par, oracle, sprite, T):
    params = T.default_params()
    aabbs, light = par.scene_synthetic(200, 480, 320, 320, 4)
    exp = oracle.render(params, aabbs, sprite, light, planes=("gbuf",))["gbuf"].reshape(320, 480)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for (x, y) in [(0, 0), (240, 160), (479, 319), (100, 37)]:
            assert r.pick(x, y).tobytes() == exp[y, x].tobytes()  # `mouse_pixel`, alt:380-382
        r.render(("fb",))
        st = r.stats()
        assert st.entities == 200 and st.bin_insertions > 0 and st.occupied_columns > 0


def test_other_bin_sizes_and_palettes(par, oracle, sprite, T):
    for bin_size, (w, h, l) in [(32, (512, 384, 256)), (64, (400, 300, 300)), (20, (320, 200, 200))]:
        params = T.default_params(w, h, l, bin_size)
        params.ambient = 0.4
        params.background = 90
        params.palette_size = 4
        for i, c in enumerate([(10, 200, 30, 7), (255, 0, 0, 255), (1, 2, 3, 4), (90, 90, 250, 0)]):
            params.palette[i] = T.Color(*c)
        aabbs, light = par.scene_synthetic(220, w, h, l, bin_size)
        exp = oracle.render(params, aabbs, sprite, light)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(ALL), exp, ALL, f"bin {bin_size} dense")
            fast = r.render(("fb", "palidx", "brightness", "gbuf"))
            assert_planes_equal(fast, exp, ("fb", "palidx", "brightness", "gbuf"), f"bin {bin_size}")


def test_tile_passes_in_strips_wide_and_clipped_bins(par, oracle, sprite, T):
    """Bins wider than two sprites and views that are no multiple of the bin, end to end: a render work item visits
    its rectangle in strips of a sprite's width (csrc/par_strips.h; the arithmetic itself is checked for every
    rectangle shape by test_strip_order_visits_every_pixel_once), and a view-clipped tile can have any width — 21
    pixels (strips of 20 + 1), 1 pixel, 20. A floor plus scattered boxes of every extent; every plane against the
    oracle, default mode and every ray traced."""
    for bin_size, (w, h, l) in [(160, (500, 360, 360)), (100, (421, 300, 300)), (60, (241, 200, 200)),
                                (120, (380, 250, 250))]:
        params = T.default_params(w, h, l, bin_size)
        rows = [(i * 20, 0, j * 20, 20, 20, 20) for i in range((w + 19) // 20) for j in range(l // 20)]
        rng = np.random.default_rng(bin_size)
        rows += [(int(rng.integers(0, w - 20)), int(rng.integers(20, 120)), int(rng.integers(0, l - 20)),
                  int(rng.integers(1, 21)), int(rng.integers(1, 21)), int(rng.integers(1, 21))) for _ in range(150)]
        aabbs = T.make_aabbs(rows)
        light = T.make_light(w // 2 + 33, h // 2, l // 4)
        exp = oracle.render(params, aabbs, sprite, light, nthreads=8)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(ALL), exp, ALL, f"bin {bin_size} every ray")
            fast = r.render(("fb", "palidx", "brightness", "gbuf"))
            assert_planes_equal(fast, exp, ("fb", "palidx", "brightness", "gbuf"), f"bin {bin_size}")


def test_full_size_headline_frame(par, oracle, sprite, T):
    """BASELINE's headline configuration itself: 4096x4096, 1024 primitives — bit-exact against the oracle (rows
    split over the host cores), plus the properties that do not depend on the oracle: the default (background rays
    skipped) and the every-ray-traced modes give the same frame, rendering is idempotent, and row blocks of any
    alignment reproduce the rows of the whole frame."""
    import os
    w = h = l = 4096
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(1024, w, h, l, 12345)
    exp = oracle.render(params, aabbs, sprite, light, nthreads=os.cpu_count() or 8, planes=("fb", "palidx"))
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        a = r.render(("fb", "palidx"))
        assert a["fb"].tobytes() == exp["fb"].tobytes() and a["palidx"].tobytes() == exp["palidx"].tobytes()
        b = r.render(("fb", "palidx"), flags=par.RENDER_TRACE_BACKGROUND)
        assert b["fb"].tobytes() == a["fb"].tobytes() and b["palidx"].tobytes() == a["palidx"].tobytes()
        c = r.render(("fb", "palidx"))
        assert c["fb"].tobytes() == a["fb"].tobytes()
        for r0, r1 in [(0, 512), (512, 1024), (3584, 4096), (1999, 2113)]:
            blk = r.render(("fb",), rows=(r0, r1))
            assert blk["fb"].tobytes() == a["fb"][r0 * w:r1 * w].tobytes(), (r0, r1)


def test_host_demo_binary(par, oracle, T, tmp_path):
    """The C++ host program (the reference's main loop on the C ABI) renders the default scene and its scripted
    frames; its PPM output equals the oracle's frames with the debug line (SURVEY Appendix B frames)."""
    import os
    import subprocess
    demo = os.path.join(os.path.dirname(par.LIB_PATH), "par_demo")
    assert os.path.exists(demo), "build with make -C pixel-art-raytracer_amd/csrc"
    p = subprocess.run([demo, "--keys", "RRRRUUUU", "--frames", "9", "--out", str(tmp_path), "--debug-line"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    params = T.default_params()
    aabbs = par.scene_graybox(480, 320)
    light = T.make_light(480, 160, 80)
    sprite = par.tile_floor()
    for f in range(9):
        if f:
            aabbs[0]["px" if f <= 4 else "pz"] += 5
        out = oracle.render(params, aabbs, sprite, light, planes=("fb", "gbuf"))
        fb = out["fb"]
        oracle.debug_line(params, out["gbuf"], light, 0, 0, fb)
        raw = open(tmp_path / f"frame_{f:03d}.ppm", "rb").read()
        header = b"P6\n480 320\n255\n"
        assert raw.startswith(header)
        rgb = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(-1, 3)
        assert np.array_equal(rgb[:, 0], fb["red"]) and np.array_equal(rgb[:, 1], fb["green"]) and \
            np.array_equal(rgb[:, 2], fb["blue"]), f"frame {f}"


def test_overflow_kernel_on_every_column(tmp_path):
    """PAR_FORCE_GENERIC=1 sends every column through render_overflow_kernel (primary pass straight from the hash,
    shadow walks in-kernel): the path overflowed columns take. Run in a fresh process (the switch is read once) and
    compare with the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import importlib, sys
sys.path.insert(0, %r)
import numpy as np
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
from oracle.oracle import Oracle
o = Oracle(); sprite = par.tile_floor()
ALL = ("fb", "gbuf", "palidx", "brightness", "lit")
for (w, h, l, n, seed) in [(480, 320, 320, 300, 5), (512, 512, 512, 64, 12345), (500, 333, 290, 200, 1)]:
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(n, w, h, l, seed)
    exp = o.render(params, aabbs, sprite, light)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for planes in (ALL, ("fb", "palidx", "brightness", "gbuf")):
            out = r.render(planes)
            for k in planes:
                assert out[k].tobytes() == exp[k].tobytes(), (w, h, k)
rng = np.random.default_rng(7)
for case in range(24):
    b = int(rng.choice([8, 16, 20, 32, 40, 40, 64]))
    w, h, l = int(rng.integers(5, 80)) * 8, int(rng.integers(40, 400)), int(rng.integers(40, 400))
    n = int(rng.integers(1, 300))
    params = T.default_params(w, h, l, b)
    aabbs, light = par.scene_synthetic(n, w, h, l, int(rng.integers(1, 1 << 30)))
    if case %% 3 == 1:
        aabbs["px"] = (aabbs["px"] %% max(2 * b, 40)).astype(aabbs["px"].dtype)
        aabbs["pz"] = (aabbs["pz"] %% max(3 * b, 60)).astype(aabbs["pz"].dtype)
    if case %% 4 == 2:
        light = T.make_light(int(rng.integers(-100, w + 100)), int(rng.integers(-100, h + 100)), int(rng.integers(-100, l + 100)))
    exp = o.render(params, aabbs, sprite, light)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for planes in (("fb", "palidx"), ALL):
            out = r.render(planes)
            for k in planes:
                assert out[k].tobytes() == exp[k].tobytes(), (case, w, h, l, b, n, k)
aab = par.scene_graybox(); light = T.make_light(480, 160, 80); params = T.default_params()
exp = o.render(params, aab, sprite, light)
with par.Renderer(params) as r:
    r.set_scene(aab, sprite, light)
    out = r.render(ALL)
    for k in ALL:
        assert out[k].tobytes() == exp[k].tobytes(), k
print("generic ok")
''' % root
    env = dict(os.environ, PAR_FORCE_GENERIC="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "generic ok" in p.stdout, p.stderr[-3000:]


def test_column_teams_of_every_size():
    """The column launch's workgroups are teams of 1, 2, 4 or 8 wavefronts per column, chosen by the frame's size and
    by whether it is one of several in flight; PAR_TUNE_COL_ROLES forces one size (read once per process: a fresh
    process per size). Every size on a crowded small view (the graybox world: walls of many occupied bins per column,
    lists that wrap), a mid-size random view, long walks to a far light and columns at the edge of a record -- every
    plane against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import importlib, sys
sys.path.insert(0, %r)
import numpy as np
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
from oracle.oracle import Oracle
o = Oracle(); sprite = par.tile_floor()
ALL = ("fb", "gbuf", "palidx", "brightness", "lit")
scenes = [(T.default_params(), par.scene_graybox(), T.make_light(480, 160, 80))]
for (w, h, l, n, seed, b) in [(640, 400, 400, 300, 3, 40), (1024, 768, 512, 400, 9, 40), (320, 200, 1600, 250, 4, 8)]:
    a, li = par.scene_synthetic(n, w, h, l, seed)
    scenes.append((T.default_params(w, h, l, b), a, li))
# one screen column crowded with boxes along z (many occupied bins, many walks, lists near the record's limits)
rows = [(200 + (i %% 2) * 10, 40, 10 + 36 * i, 20, 20, 20) for i in range(30)] + [(100, 100, 100, 20, 20, 20)]
scenes.append((T.default_params(480, 320, 1200), T.make_aabbs(rows), T.make_light(470, 300, 1100)))
for params, aabbs, light in scenes:
    exp = o.render(params, aabbs, sprite, light, nthreads=8)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for planes in (("fb", "palidx"), ALL):
            out = r.render(planes)
            for k in planes:
                assert out[k].tobytes() == exp[k].tobytes(), (params.width, params.height, k)
print("teams ok")
''' % root
    for roles in ("1", "2", "4", "8"):
        env = dict(os.environ, PAR_TUNE_COL_ROLES=roles)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "teams ok" in p.stdout, (roles, p.stderr[-3000:])


def test_every_ray_traced_mode(par, oracle, sprite, T):
    """PAR_RENDER_TRACE_BACKGROUND without a lit plane, and the lit plane at full size: background rays included,
    the lit mask equals the oracle's (trace_hash_for_light for every pixel, alt:703-742)."""
    import os
    w, h, l = 2048, 2048, 2048
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(256, w, h, l, 12345)
    # a wall of boxes between most of the background and the light, so that many background rays are blocked
    wall = T.make_aabbs([(1200 + 20 * (i % 3), 20 * j, 20 * k, 20, 20, 20) for i in range(3) for j in range(40)
                         for k in range(0, 100, 2)])
    aabbs = np.concatenate([aabbs, wall])
    exp = oracle.render(params, aabbs, sprite, light, nthreads=os.cpu_count() or 8, planes=("fb", "lit"))
    assert 0.02 < 1.0 - exp["lit"].mean() < 0.98  # both lit and shadowed pixels exist
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        out = r.render(("fb", "lit"))
        assert out["lit"].tobytes() == exp["lit"].tobytes()
        assert out["fb"].tobytes() == exp["fb"].tobytes()
        out2 = r.render(("fb",), flags=par.RENDER_TRACE_BACKGROUND)
        assert out2["fb"].tobytes() == exp["fb"].tobytes()


def test_edge_views_and_lights(par, oracle, sprite, T):
    """Odd view sizes (width not a multiple of 8: the generic fill path; views smaller than a bin), empty scenes,
    lights outside the view volume in every direction (out-of-range and aliased flat bin indices, SURVEY a-4),
    a light inside a primitive, a light in the start bin (zero-length walk)."""
    cases = [
        (123, 77, 91, 60, 1, (60, 40, 20)),
        (37, 29, 33, 25, 2, (10, 10, 10)),          # smaller than one bin
        (481, 321, 321, 200, 3, (481, 160, 80)),    # light bin-x == grid width (the reference's default case)
        (480, 320, 320, 300, 4, (-300, 500, -200)),
        (480, 320, 320, 300, 5, (900, -400, 700)),
        (480, 320, 320, 300, 6, (240, 160, 5000)),
        (480, 320, 320, 0, 7, (240, 160, 80)),      # empty scene
        (640, 200, 400, 250, 8, (0, 0, 0)),
    ]
    for (w, h, l, n, seed, lpos) in cases:
        params = T.default_params(w, h, l)
        aabbs, _ = par.scene_synthetic(n, w, h, l, seed)
        light = T.make_light(*lpos)
        exp = oracle.render(params, aabbs, sprite, light)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(ALL), exp, ALL, f"{w}x{h} light {lpos}")
            fast = r.render(("fb", "palidx", "brightness", "gbuf"))
            assert_planes_equal(fast, exp, ("fb", "palidx", "brightness", "gbuf"), f"{w}x{h} light {lpos} fast")
    # light exactly on primitives' planes and in their bins: zero light-vector components -> inf/NaN slabs (a-5)
    params = T.default_params()
    rows = [(i * 20, 0, j * 20, 20, 20, 20) for i in range(24) for j in range(16)]
    rows += [(200, 20, 100, 20, 20, 20), (220, 40, 100, 20, 20, 20), (200, 20, 140, 20, 20, 20)]
    aabbs = T.make_aabbs(rows)
    for lpos in [(210, 40, 110), (200, 20, 100), (240, 20, 100), (210, 30, 110), (0, 20, 0)]:
        light = T.make_light(*lpos)
        exp = oracle.render(params, aabbs, sprite, light)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(ALL), exp, ALL, f"light on planes {lpos}")


def test_config5_scale_animation(par, oracle, sprite, T):
    """BASELINE config 5 at its own size: 1024x1024, 512 moving primitives, hipGraph replay per frame with the frames
    IN FLIGHT (no host wait between stage and launch: a graph's copy nodes read its staging area when the graph runs,
    so frame f + 1 must not be staged over frame f). Every frame is copied out in stream order and compared with a
    second renderer's blocking render of the same scene; every 10th also with the oracle. Positions keep moving by
    the reference's step of 5 (alt:643-678); the light moves through par_set_light as well as par_graph_stage."""
    import os
    import torch
    w = h = l = 1024
    n = 512
    frames = 60
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(n, w, h, l, 99)
    rng = np.random.default_rng(11)
    vel = rng.choice([-5, 0, 5], size=(n, 3)).astype(np.int16)
    fb = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    ring = torch.zeros(frames, w * h * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    scenes = []
    with par.Renderer(params) as r, par.Renderer(params) as check:
        r.set_scene(aabbs, sprite, light)
        check.set_scene(aabbs, sprite, light)
        r.graph_capture({"fb": fb.data_ptr()}, stream=stream.cuda_stream)
        for f in range(frames):
            if f:
                aabbs["px"] += vel[:, 0]
                aabbs["py"] += vel[:, 1]
                aabbs["pz"] += vel[:, 2]
                if f % 7 == 0:
                    light["z"] += 5
                if f % 14 == 0:
                    r.graph_stage(aabbs, 0)
                    r.set_light(light)       # the light alone, outside the stage call
                else:
                    r.graph_stage(aabbs, 0, light)
            r.graph_launch(stream.cuda_stream)
            with torch.cuda.stream(stream):
                ring[f].copy_(fb, non_blocking=True)
            scenes.append((aabbs.copy(), light.copy()))
        stream.synchronize()
        got = ring.cpu().numpy()
        for f, (a, li) in enumerate(scenes):
            check.update_aabbs(a, 0)
            check.set_light(li)
            exp = check.render(("fb",))["fb"].view(np.uint8)
            assert np.array_equal(got[f], exp), f"frame {f}: graph replay with frames in flight"
            if f % 10 == 0 or f == frames - 1:
                ora = oracle.render(params, a, sprite, li, nthreads=os.cpu_count() or 8, planes=("fb",))
                assert np.array_equal(got[f], ora["fb"].view(np.uint8)), f"frame {f} vs oracle"


def test_config5_full_run_frames_in_flight(par, oracle, sprite, T):
    """BASELINE config 5 for its full 300 frames through the frames-in-flight path: four slots, every slot's context
    gets the frame's AABBs in stream order (par_update_aabbs_async) before it renders; every 25th frame (and the
    last) is compared with the oracle."""
    import importlib
    import os
    import torch
    pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")
    w = h = l = 1024
    n = 512
    params = T.default_params(w, h, l)
    aabbs0, light = par.scene_synthetic(n, w, h, l, 77)
    rng = np.random.default_rng(5)
    vel = rng.choice([-5, 0, 5], size=(n, 3)).astype(np.int16)
    pipe = pipeline.FramePipeline(params, aabbs0, sprite, light, depth=4, planes=("fb", "palidx"), calibrate=False)
    try:
        for f in range(300):
            a = aabbs0.copy()
            a["px"] += vel[:, 0] * f
            a["py"] += vel[:, 1] * f
            a["pz"] += vel[:, 2] * f
            pipe.update_aabbs(f, a)
            slot = pipe.submit(f)
            if f % 25 == 0 or f == 299:
                slot.stream.synchronize()
                exp = oracle.render(params, a, sprite, light, nthreads=os.cpu_count() or 8, planes=("fb", "palidx"))
                assert np.array_equal(slot.buffers["fb"].cpu().numpy(), exp["fb"].view(np.uint8)), f"frame {f}"
                assert np.array_equal(slot.buffers["palidx"].cpu().numpy(), exp["palidx"]), f"frame {f}"
    finally:
        pipe.close()


def test_graph_is_dropped_with_its_sprite_table_and_reports_allocation_failure(par, sprite, T, monkeypatch):
    """par_set_sprites frees the tables a captured graph's kernels point at: the graph must be gone afterwards
    (PAR_ERR_NOT_READY, not a read of freed memory). And a host allocation failure inside the library comes back as
    PAR_ERR_OOM through the C ABI, never as an exception (PAR_TEST_BAD_ALLOC=1 makes the guarded bodies throw)."""
    import torch
    params = T.default_params(256, 256, 256)
    aabbs, light = par.scene_synthetic(40, 256, 256, 256, 3)
    fb = torch.zeros(256 * 256 * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        r.graph_capture({"fb": fb.data_ptr()}, stream=stream.cuda_stream)
        r.graph_launch(stream.cuda_stream)
        stream.synchronize()
        r.set_sprites(sprite)
        with pytest.raises(par.ParError) as e:
            r.graph_launch(stream.cuda_stream)
        assert e.value.status == 8  # PAR_ERR_NOT_READY
        monkeypatch.setenv("PAR_TEST_BAD_ALLOC", "1")
        with pytest.raises(par.ParError) as e:
            r.set_entities(aabbs)
        assert e.value.status == 4  # PAR_ERR_OOM
        with pytest.raises(par.ParError) as e:
            r.set_sprites(sprite)
        assert e.value.status == 4
        monkeypatch.delenv("PAR_TEST_BAD_ALLOC")
        r.set_scene(aabbs, sprite, light)  # the context is still usable
        assert r.render(("fb",))["fb"].shape[0] == 256 * 256


def test_config1_default_scene_128(par, oracle, sprite, T):
    """BASELINE config 1: 128x128 view of the reference's default graybox world (alt:517-599 parameterised on the
    view), light at (W, H/2, L/4) as alt:625-626 (its bin-x equals the grid width: the out-of-range case)."""
    params = T.default_params(128, 128, 128)
    aabbs = par.scene_graybox(128, 128)
    light = T.make_light(128, 64, 32)
    exp = oracle.render(params, aabbs, sprite, light)
    assert (exp["palidx"] != T.PALIDX_BACKGROUND).mean() > 0.5
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        assert_planes_equal(r.render(ALL), exp, ALL, "128x128 graybox")


@pytest.mark.parametrize("world,assemble", [(2, "tiles"), (3, "tiles"), (3, "blocks"), (2, "none")])
def test_bench_multi_rank_control_flow(world, assemble):
    """bench.py's N > 1 path (row blocks, frames in flight, per-frame gather, verification of the assembled frame)
    with `world` ranks sharing the one GPU of the test box. RCCL cannot run several ranks on one device, so the
    exchange goes through gloo here; everything else is the code the 8-GPU run executes."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", str(world), "--steps", "12", "--warmup", "3", "--blocks", "3", "--backend", "gloo", "--share-gpu",
           "--size", "1000" if world == 3 else "1024", "--no-cpu-baseline", "--assemble", assemble]
    # PAR_BENCH_RAMP_SKEW: the ranks' clocks WANT different numbers of untimed ramp iterations (each holds
    # collectives); the job only ends if they agree on one number all the same
    p = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PAR_BENCH_RAMP_SKEW="0.25"))
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["steps"] == 12 and d["verified_vs_single_gpu_frame"] is True
    assert d["value"] > 0 and d["scaling"] == "strong"
    ramp = d["multi_gpu"]["ramp_iterations"]
    assert len(ramp) == world and len(set(ramp)) == 1 and ramp[0] >= 1, ramp  # every rank ran the same number
    mg = d["multi_gpu"]
    assert len(mg["ranks_seen"]) == world and mg["render_ms"] > 0 and mg["assemble"] == assemble
    assert d["ms_per_step_spread"]["blocks"] == 3
    if assemble == "none":
        assert mg["gather_ms"] == 0 and mg["gather_bytes_per_rank"] == 0
    else:
        assert mg["gather_ms"] > 0 and mg["gather_bytes_per_rank"] > 0
    if assemble == "tiles":
        t = mg["tiles"]
        assert 0 < t["total"] <= t["of"] and sum(t["per_rank"]) == t["total"] and len(t["per_rank"]) == world
        assert t["bytes_to_rank0"] < 4 * d_size(d) ** 2  # less than the whole frame travels


def d_size(d):
    """The view size of a bench line (its metric names it)."""
    return int(d["metric"].split(" at ")[1].split("x")[0])


def test_eight_row_blocks_and_their_tiles_assemble_the_headline_frame(par, oracle, sprite, T):
    """The partition an 8-GPU run of the headline uses, on one GPU: the eight blocks par_row_block(r, 8, 4096, 40)
    rendered one by one (a) concatenate to the whole-frame render, which equals the oracle; (b) packed into the tiles
    that can show a primitive (par_scene_tiles / par_tiles_pack) and assembled on a background-filled frame
    (par_background_fill / par_tiles_unpack) give the same frame; (c) with rank 0's block rendered straight into its
    rows of the frame and every other row written once from the tile map (par_scene_tile_map / par_tiles_assemble):
    the same frame again -- what rank 0 of the sharded run does."""
    import importlib
    import torch
    sharding = importlib.import_module("pixel-art-raytracer_amd.sharding")
    w = h = l = 4096
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(1024, w, h, l, 12345)
    blocks = [sharding.row_block(r, 8, h, params.bin_size) for r in range(8)]
    assert blocks == [(0, 480), (480, 1000), (1000, 1520), (1520, 2040), (2040, 2560), (2560, 3080), (3080, 3600), (3600, 4096)]
    tiles = par.scene_tiles(params, aabbs)
    assert 0 < len(tiles) < np.prod(params.grid_dims()[:2]) // 2     # a sparse frame: most tiles never travel
    dev = torch.device("cuda", 0)
    d_tiles = torch.from_numpy(tiles.copy()).to(dev)
    slot = params.bin_size * params.bin_size * 4
    inbox = torch.zeros(len(tiles) * slot, dtype=torch.uint8, device=dev)
    frame = torch.full((h * w * 4,), 7, dtype=torch.uint8, device=dev)
    by = tiles >> 16
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        whole = r.render(("fb", "palidx"))
        parts = []
        for (b, e) in blocks:
            blk = torch.zeros((e - b) * w * 4, dtype=torch.uint8, device=dev)
            r.render_device({"fb": blk.data_ptr()}, rows=(b, e))
            torch.cuda.synchronize()
            parts.append(blk.cpu().numpy())
            first = int(np.searchsorted(by, b // params.bin_size)); end = int(np.searchsorted(by, -(-e // params.bin_size)))
            par.tiles_pack(params, d_tiles.data_ptr() + 4 * first, end - first, blk.data_ptr(), (b, e),
                           inbox.data_ptr() + first * slot)
        par.background_fill(params, frame.data_ptr(), h)
        par.tiles_unpack(params, d_tiles.data_ptr(), len(tiles), inbox.data_ptr(), frame.data_ptr())
        torch.cuda.synchronize()
        # (c) in place, for the first, a middle and the last block as the assembling rank's own
        d_map = torch.from_numpy(par.scene_tile_map(params, tiles)).to(dev)
        in_place = []
        for root in (0, 3, 7):
            f2 = torch.full((h * w * 4,), 9, dtype=torch.uint8, device=dev)
            b, e = blocks[root]
            r.render_device({"fb": f2.data_ptr() + b * w * 4}, rows=(b, e))
            par.tiles_assemble(params, d_map.data_ptr(), inbox.data_ptr(), f2.data_ptr(), (0, b))
            par.tiles_assemble(params, d_map.data_ptr(), inbox.data_ptr(), f2.data_ptr(), (e, h))
            torch.cuda.synchronize()
            in_place.append(f2.cpu().numpy())
    whole_fb = whole["fb"].view(np.uint8)
    assert np.array_equal(np.concatenate(parts), whole_fb)
    assert np.array_equal(frame.cpu().numpy(), whole_fb)
    for f2 in in_place:
        assert np.array_equal(f2, whole_fb)
    exp = oracle.render(params, aabbs, sprite, light, nthreads=os.cpu_count() or 8, planes=("fb", "palidx"))
    assert whole["fb"].tobytes() == exp["fb"].tobytes() and whole["palidx"].tobytes() == exp["palidx"].tobytes()


@pytest.mark.parametrize("w,h,b", [(333, 170, 16), (250, 200, 10), (480, 320, 40)])
def test_tiles_assemble_on_views_of_any_width(par, sprite, T, w, h, b):
    """par_tiles_assemble where the 16-byte path does not apply (width or bin size not a multiple of 4) and where it
    does: three ranks' blocks packed, the root's (rank 1) rendered in place, the other rows assembled in one pass."""
    import importlib
    import torch
    sharding = importlib.import_module("pixel-art-raytracer_amd.sharding")
    params = T.default_params(w, h, 120, b)
    aabbs, light = par.scene_synthetic(90, w, h, 120, 31)
    dev = torch.device("cuda", 0)
    gathers = [sharding.TileGather(params, aabbs, dev, world=3, rank=q, dst=1, in_place=True) for q in range(3)]
    root = gathers[1]
    slot = root.slot_bytes
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        whole = r.render(("fb",))["fb"].view(np.uint8)
        for q, g in enumerate(gathers):
            rows = g.blocks[q]
            if rows[1] <= rows[0]:
                continue
            if q == 1:
                r.render_device({"fb": root.root_block().data_ptr()}, rows=rows)
                continue
            blk, packed = g.block_buffer(), g.packed_buffer()
            r.render_device({"fb": blk.data_ptr()}, rows=rows)
            g.pack(blk, packed)
            n = g.counts[q]
            root.inbox[g.first[q] * slot:(g.first[q] + n) * slot] = packed[:n * slot]  # (what the send would do)
        root.assemble()
        torch.cuda.synchronize()
    assert np.array_equal(root.frame.cpu().numpy(), whole)


def test_host_demo_gif(par, oracle, T, tmp_path):
    """The animated GIF the host program writes (frame sink replacing the SDL present): every frame decodes to
    exactly the oracle's frame (each frame's colours fit one exact 256-entry local colour table)."""
    import os
    import subprocess
    demo = os.path.join(os.path.dirname(par.LIB_PATH), "par_demo")
    gif = tmp_path / "anim.gif"
    p = subprocess.run([demo, "--keys", "RRUUhhjjPP", "--frames", "6", "--gif", str(gif), "--debug-line"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    frames = _decode_gif(gif.read_bytes())
    assert len(frames) == 6
    params = T.default_params()
    aabbs = par.scene_graybox(480, 320)
    light = T.make_light(480, 160, 80)
    sprite = par.tile_floor()
    from helpers import apply_key
    for f in range(6):
        if f:
            apply_key("RRUUhhjjPP"[f - 1], aabbs, light)
        out = oracle.render(params, aabbs, sprite, light, planes=("fb", "gbuf"))
        fb = out["fb"]
        oracle.debug_line(params, out["gbuf"], light, 0, 0, fb)
        exp = np.stack([fb["red"], fb["green"], fb["blue"]], axis=1)
        assert np.array_equal(frames[f], exp), f"frame {f}"


def _decode_gif(data):
    """Minimal GIF89a decoder (local colour tables, no interlace): list of (H*W, 3) uint8 arrays."""
    assert data[:6] == b"GIF89a"
    w, h = int.from_bytes(data[6:8], "little"), int.from_bytes(data[8:10], "little")
    pos = 13
    if data[10] & 0x80:
        pos += 3 * (2 << (data[10] & 7))
    frames = []
    while True:
        b = data[pos]
        if b == 0x3B:
            break
        if b == 0x21:  # extension: skip sub-blocks
            pos += 2
            while data[pos]:
                pos += 1 + data[pos]
            pos += 1
            continue
        assert b == 0x2C
        fw, fh = int.from_bytes(data[pos + 5:pos + 7], "little"), int.from_bytes(data[pos + 7:pos + 9], "little")
        flags = data[pos + 9]
        pos += 10
        assert (fw, fh) == (w, h) and (flags & 0x80) and not (flags & 0x40)
        n = 2 << (flags & 7)
        table = np.frombuffer(data[pos:pos + 3 * n], dtype=np.uint8).reshape(n, 3)
        pos += 3 * n
        min_bits = data[pos]
        pos += 1
        stream = bytearray()
        while data[pos]:
            stream += data[pos + 1:pos + 1 + data[pos]]
            pos += 1 + data[pos]
        pos += 1
        # LZW
        clear, eoi = 1 << min_bits, (1 << min_bits) + 1
        bits, nxt = min_bits + 1, eoi + 1
        dict_ = {i: bytes([i]) for i in range(clear)}
        out = bytearray()
        acc = nacc = 0
        prev = None
        it = iter(stream)
        done = False
        while not done:
            while nacc < bits:
                try:
                    acc |= next(it) << nacc
                except StopIteration:
                    done = True
                    break
                nacc += 8
            if done:
                break
            code = acc & ((1 << bits) - 1)
            acc >>= bits
            nacc -= bits
            if code == clear:
                dict_ = {i: bytes([i]) for i in range(clear)}
                bits, nxt, prev = min_bits + 1, eoi + 1, None
                continue
            if code == eoi:
                break
            if prev is None:
                entry = dict_[code]
            else:
                entry = dict_[code] if code in dict_ else prev + prev[:1]
                if nxt < 4096:
                    dict_[nxt] = prev + entry[:1]
                    nxt += 1
                    if nxt == (1 << bits) and bits < 12:
                        bits += 1
            out += entry
            prev = entry
        idx = np.frombuffer(bytes(out[:w * h]), dtype=np.uint8)
        assert len(idx) == w * h
        frames.append(table[idx])
    return frames


def test_deep_grid_and_far_light(par, oracle, sprite, T):
    """A grid 1000 bins deep (the per-column scans iterate) and lights thousands of bins away (walks far longer
    than one staging pass; flat bin indices far outside the grid)."""
    w, h, l, b = 96, 80, 8000, 8
    params = T.default_params(w, h, l, b)
    assert params.grid_dims() == (12, 10, 1000)
    rng = np.random.default_rng(3)
    n = 400
    aabbs = np.zeros(n, dtype=T.AABB)
    aabbs["px"] = rng.integers(-10, w, n)
    aabbs["py"] = rng.integers(-30, 60, n)
    aabbs["pz"] = rng.integers(-20, l, n)
    aabbs["pz"][:150] = rng.integers(0, 120, 150)  # many near the front so that pixels are covered
    aabbs["ex"] = rng.integers(1, 21, n)
    aabbs["ey"] = rng.integers(0, 21, n)
    aabbs["ez"] = rng.integers(0, 21, n)
    for lpos in [(48, 40, 30000), (48, 30000, 60), (-20000, 10, 10), (50, 20, 4000)]:
        light = T.make_light(*lpos)
        exp = oracle.render(params, aabbs, sprite, light)
        assert (exp["palidx"] != T.PALIDX_BACKGROUND).any()
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(ALL), exp, ALL, f"deep grid light {lpos}")
            fast = r.render(("fb", "palidx", "brightness", "gbuf"))
            assert_planes_equal(fast, exp, ("fb", "palidx", "brightness", "gbuf"), f"deep grid light {lpos} fast")


def test_dense_floor_full_size(par, oracle, sprite, T):
    """4096x4096 covered by a full floor of 41 943 tiles (83 028 bin insertions: the node pool grows past its
    initial size; every column is occupied; most columns are visited as whole tiles)."""
    import os
    w = h = l = 4096
    params = T.default_params(w, h, l)
    aabbs = T.make_aabbs([(i * 20, 0, j * 20, 20, 20, 20) for i in range(w // 20) for j in range(l // 20)])
    light = T.make_light(2560, 2048, 1024)
    exp = oracle.render(params, aabbs, sprite, light, nthreads=os.cpu_count() or 8, planes=("fb", "palidx"))
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        out = r.render(("fb", "palidx"))
        assert r.stats().bin_insertions > 65536
        assert out["fb"].tobytes() == exp["fb"].tobytes() and out["palidx"].tobytes() == exp["palidx"].tobytes()


def test_random_sweep(par, oracle, sprite, T):
    """Many small random configurations through the C ABI against the oracle: view sizes that are no multiple of the
    bin size, bin sizes 8..64, lights inside, on the border of and outside the volume, clumped and spread primitives, odd extents,
    row ranges, the riding fill (frame + palette index) and the standalone fill paths (every plane)."""
    rng = np.random.default_rng(20260104)
    for case in range(96):
        b = int(rng.choice([8, 16, 20, 24, 32, 40, 40, 40, 48, 64]))
        w = int(rng.integers(5, 90)) * 8 if case % 3 else int(rng.integers(40, 700))
        h = int(rng.integers(40, 500))
        l = int(rng.integers(40, 500))
        n = int(rng.integers(1, 400))
        params = T.default_params(w, h, l, b)
        aabbs, light = par.scene_synthetic(n, w, h, l, int(rng.integers(1, 1 << 30)))
        if case % 4 == 1:  # clump: many primitives in few bins (wrapping slot counters, long entry lists)
            aabbs["px"] = (aabbs["px"] % max(2 * b, 40)).astype(aabbs["px"].dtype)
            aabbs["pz"] = (aabbs["pz"] % max(3 * b, 60)).astype(aabbs["pz"].dtype)
        if case % 6 == 3:  # any extents the sprite format allows (alt:330: ex <= 20, ey + ez <= 40), also 1 and 0 wide
            aabbs["ex"] = rng.integers(0, 21, n).astype(aabbs["ex"].dtype)
            aabbs["ey"] = rng.integers(0, 21, n).astype(aabbs["ey"].dtype)
            aabbs["ez"] = (rng.integers(0, 21, n) % (41 - aabbs["ey"])).astype(aabbs["ez"].dtype)
        if case % 5 == 2:  # light anywhere, also outside the volume and with negative coordinates
            light = T.make_light(int(rng.integers(-100, w + 100)), int(rng.integers(-100, h + 100)),
                                 int(rng.integers(-100, l + 100)))
        exp = oracle.render(params, aabbs, sprite, light)
        tag = f"case {case}: {w}x{h}x{l} bin {b}, {n} primitives, light {light}"
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(("fb", "palidx")), exp, ("fb", "palidx"), tag + " (riding fill)")
            assert_planes_equal(r.render(ALL), exp, ALL, tag + " (all planes)")
            r0 = int(rng.integers(0, h - 1))
            r1 = int(rng.integers(r0 + 1, h + 1))
            part = r.render(("fb", "palidx", "lit"), rows=(r0, r1))
            for k in ("fb", "palidx", "lit"):
                assert part[k].tobytes() == exp[k][r0 * w:r1 * w].tobytes(), tag + f" rows {r0}..{r1} plane {k}"


def test_async_updates_with_frames_in_flight(par, oracle, sprite, T):
    """Moving primitives with four frames in flight: every slot's context gets the frame's AABBs in stream order
    (par_update_aabbs_async, no host wait) before it renders; every frame equals the oracle's."""
    import importlib
    import torch
    pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")
    w, h, l = 640, 480, 400
    params = T.default_params(w, h, l)
    n = 300
    aabbs0, light = par.scene_synthetic(n, w, h, l, 17)
    rng = np.random.default_rng(9)
    vel = rng.choice([-5, 0, 5], size=(n, 3)).astype(np.int16)
    frames = 24

    def scene(f):
        a = aabbs0.copy()
        a["px"] += vel[:, 0] * f
        a["py"] += vel[:, 1] * f
        a["pz"] += vel[:, 2] * f
        return a

    pipe = pipeline.FramePipeline(params, aabbs0, sprite, light, depth=4)
    got = []
    try:
        for f in range(frames):
            slot = pipe.slot(f)
            if f >= 4:  # the slot's previous frame: read it back before the slot is reused
                slot.stream.synchronize()
                got.append((f - 4, slot.buffers["fb"].cpu().numpy().copy(), slot.buffers["palidx"].cpu().numpy().copy()))
            pipe.update_aabbs(f, scene(f))
            pipe.submit(f)
        pipe.synchronize()
        for f in range(frames - 4, frames):
            slot = pipe.slot(f)
            got.append((f, slot.buffers["fb"].cpu().numpy().copy(), slot.buffers["palidx"].cpu().numpy().copy()))
    finally:
        pipe.close()
    assert len(got) == frames
    for f, fb, pal in got:
        exp = oracle.render(params, scene(f), sprite, light, planes=("fb", "palidx"))
        assert np.array_equal(fb, exp["fb"].view(np.uint8)), f"frame {f}"
        assert np.array_equal(pal, exp["palidx"]), f"frame {f}"


def test_cpp_pipeline_host(par):
    """The frames-in-flight loop in host C++ (par_pipeline): four contexts/streams, moving primitives sent with
    par_update_aabbs_async; its last frames equal the blocking render of the same scenes (--check)."""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(par.LIB_PATH), "par_pipeline")
    assert os.path.exists(exe), "build with make -C pixel-art-raytracer_amd/csrc"
    for extra in (["--size", "1024", "--prims", "512", "--frames", "120", "--moving"],
                  ["--size", "640", "--prims", "100", "--frames", "60", "--inflight", "3"]):
        p = subprocess.run([exe, "--check"] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "check: ok" in p.stdout, p.stdout + p.stderr
        line = json.loads(p.stdout.splitlines()[0])
        assert line["host"] == "C++" and line["frames_per_s"] > 0


@pytest.mark.parametrize("gather", ["tiles", "tiles-copy", "blocks", "none"])
def test_cpp_ranks_host_gathers_over_rccl(par, tmp_path, gather):
    """The sharded-frame loop in host C++ (par_ranks): one process per GPU, row blocks cut at bin rows, the frame's
    exchange enqueued behind the render on the frame's stream: the tiles that can show a primitive to rank 0, which
    writes the background itself (tiles: rank 0's own block rendered in place in the assembled frame; tiles-copy: its
    block packed and unpacked like everyone's), ONE ncclGather of the blocks (blocks), or nothing (none). A one-GPU box can
    run it with one rank, which still executes the whole per-frame path: communicator set-up from the id file, render
    into the block, pack / background / unpack (or RCCL's gather of the only block), assembled frame equal to the
    whole-frame render (--check)."""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(par.LIB_PATH), "par_ranks")
    assert os.path.exists(exe), "build with make -C pixel-art-raytracer_amd/csrc"
    p = subprocess.run([exe, "--ranks", "1", "--rank", "0", "--id-file", str(tmp_path / "rccl_ids"), "--size", "1024",
                        "--prims", "300", "--frames", "80", "--inflight", "3", "--gather", gather, "--check"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and f"check (rank 0, gather {gather}): ok" in p.stdout, p.stdout + p.stderr
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["host"] == "C++ ranks" and line["ranks"] == 1 and line["gather"] == gather
    assert line["bytes_to_rank0_per_frame"] == 0  # (one rank: nothing travels)
    if gather in ("tiles", "tiles-copy"):
        assert 0 < line["tiles"] < 26 * 26


def test_one_launch_hash_build_equals_two_launches(par, oracle, sprite, T):
    """Small scenes build the spatial hash in ONE launch (insert, a barrier among the build workgroups, resolve)
    instead of two, and the work items of columns that hold one entity and cast no shadow on themselves carry all the
    render kernel needs (it never reads their record). A moving scene, 120 frames with the riding fill in the same
    launches: every frame equals the same frame built with two launches (flag bit 23) and rendered through the
    records only (flag bit 22), and every 20th the oracle's."""
    import torch
    w, h, l = 1024, 768, 640
    params = T.default_params(w, h, l)
    n = 700
    aabbs, light = par.scene_synthetic(n, w, h, l, 77)
    rng = np.random.default_rng(11)
    vel = rng.choice([-5, 0, 5], size=(n, 3)).astype(np.int16)
    bufs = [{k: torch.zeros(w * h * (4 if k == "fb" else 1), dtype=torch.uint8, device="cuda")
             for k in ("fb", "palidx")} for _ in range(3)]
    ptrs = [{k: v.data_ptr() for k, v in b.items()} for b in bufs]
    stream = torch.cuda.Stream()
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for f in range(120):
            aabbs["px"] += vel[:, 0]
            aabbs["py"] += vel[:, 1]
            aabbs["pz"] += vel[:, 2]
            r.update_aabbs(aabbs, 0, stream=stream.cuda_stream)
            r.render_device(ptrs[0], stream=stream.cuda_stream)                  # one launch
            r.render_device(ptrs[1], stream=stream.cuda_stream, flags=1 << 23)   # two launches
            r.render_device(ptrs[2], stream=stream.cuda_stream, flags=1 << 22)   # every work item through its record
            stream.synchronize()
            for k in ("fb", "palidx"):
                assert torch.equal(bufs[0][k], bufs[1][k]), f"frame {f} plane {k}: one-launch build differs"
                assert torch.equal(bufs[0][k], bufs[2][k]), f"frame {f} plane {k}: self-contained work items differ"
            if f % 20 == 0:
                exp = oracle.render(params, aabbs, sprite, light, planes=("fb", "palidx"), nthreads=8)
                assert np.array_equal(bufs[0]["fb"].cpu().numpy(), exp["fb"].view(np.uint8)), f"frame {f}"
                assert np.array_equal(bufs[0]["palidx"].cpu().numpy(), exp["palidx"]), f"frame {f}"


def test_device_failure_is_reported_through_the_abi():
    """A kernel-side failure surfaces as PAR_ERR_DEVICE, once, from the first call that waits for the device.
    PAR_TEST_LOSE_BUILD_WG=1 (read once per process: fresh process) makes build workgroup 0 of the one-launch hash build
    never arrive at its barrier; the others give up after their (shortened) bound, skip resolve and set the sticky
    error word. The frame is then empty, and every entry point that synchronises says so."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import importlib, sys
sys.path.insert(0, %r)
import numpy as np, torch
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
params = T.default_params(512, 512, 512)
aabbs, light = par.scene_synthetic(64, 512, 512, 512, 12345)
with par.Renderer(params) as r:
    r.set_scene(aabbs, par.tile_floor(), light)
    try:
        r.render(("fb",))
        raise SystemExit("par_render did not report the lost workgroup")
    except par.ParError as e:
        assert e.status == 9, e.status       # PAR_ERR_DEVICE
    # the asynchronous entry point cannot see it; the next call that waits does, once
    fb = torch.zeros(512 * 512 * 4, dtype=torch.uint8, device="cuda")
    r.render_device({"fb": fb.data_ptr()}, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    try:
        r.stats()
        raise SystemExit("par_get_stats did not report the lost workgroup")
    except par.ParError as e:
        assert e.status == 9, e.status
    st = r.stats()                            # reported once: the flag is clear again, no frame since
    assert st.occupied_columns == 0           # (resolve was skipped: the failed frame is visibly empty)
    assert not fb.cpu().numpy().reshape(-1, 4)[:, 3].any() and (fb.cpu().numpy().reshape(-1, 4)[:, 0] == 31).all()
print("device error ok")
''' % root
    env = dict(os.environ, PAR_TEST_LOSE_BUILD_WG="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "device error ok" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


@pytest.mark.parametrize("pairs", [30, 31, 32, 33, 34])
def test_columns_at_the_edge_of_a_record(par, oracle, sprite, T, pairs):
    """One screen column with `pairs` (entity, bin) pairs, around what the host takes as the sure capacity of a column
    record (32): up to there the frame is enqueued WITHOUT a launch for the overflow list, beyond it with one. Either
    way the picture is the oracle's, no column is left on an overflow list nobody renders (the device would flag
    that: PAR_ERR_DEVICE from stats()), and overflow_columns is 0 whenever the launch was skipped."""
    w, h, l = 480, 320, 1400
    params = T.default_params(w, h, l)
    # boxes with y + z constant stay on the same screen rows: one box per z-bin of screen column (2, 5)
    rows = [(90 + (b % 7), 190 - 40 * b, 40 * b + 10, 20, 20, 20) for b in range(pairs)]
    rows += [(i * 20, 0, j * 20, 20, 20, 20) for i in range(10, 24) for j in range(0, 16)]
    aabbs = T.make_aabbs(rows)
    light = T.make_light(300, 160, 80)
    exp = oracle.render(params, aabbs, sprite, light)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        planes = ("fb", "palidx", "brightness", "gbuf")
        assert_planes_equal(r.render(planes), exp, planes, f"{pairs} pairs")
        st = r.stats()  # (raises PAR_ERR_DEVICE if a column overflowed with no launch for the list)
        if pairs <= 32:
            assert st.overflow_columns == 0
