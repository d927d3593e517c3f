"""The CPU oracle against the committed golden vectors (generated from the reference's own functions by
tests/golden/make_golden.py) and the survey's whole-program known answers."""
import os

import numpy as np

from helpers import apply_key, graybox, sha, visible_hash
from oracle.oracle import GridArrays

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_struct_sizes(T):
    # SURVEY §8 a-1 [probe] sizes
    assert T.COLOR.itemsize == 4 and T.VEC3.itemsize == 12 and T.PIXEL.itemsize == 28
    assert T.SPRITE.itemsize == 16000 and T.AABB.itemsize == 16 and T.LIGHT.itemsize == 8 and T.RAY.itemsize == 20
    assert T.PIXEL.fields["color"][1] == 12 and T.PIXEL.fields["y"][1] == 16 and T.PIXEL.fields["entity_index"][1] == 24
    assert T.SPRITE.fields["depth"][1] == 3200 and T.SPRITE.fields["normal"][1] == 6400


def test_units_intersect(oracle, T):
    z = np.load(os.path.join(GOLDEN, "ref_units.npz"))
    boxes = np.ascontiguousarray(z["boxes"]).view(T.AABB).reshape(-1)
    rays = np.ascontiguousarray(z["rays"]).view(T.RAY).reshape(-1)
    got = np.array([oracle.intersect(boxes[i:i + 1], rays[i:i + 1]) for i in range(len(boxes))], dtype=np.uint8)
    assert np.array_equal(got, z["hit"])
    assert 0 < z["hit"].sum() < len(boxes)
    assert np.isnan(rays["inv_x"]).any() and np.isinf(rays["inv_y"]).any()


def test_units_color_scale(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_units.npz"))
    for row, exp in zip(z["cs_in"], z["cs_out"]):
        assert oracle.color_scale(tuple(int(c) for c in row[:4]), float(row[4])) == tuple(int(c) for c in exp)


def test_units_normalize(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_units.npz"))
    for v, exp in zip(z["nv"], z["nv_out"]):
        got = oracle.normalize(v)
        assert got.tobytes() == exp.astype(np.float32).tobytes()  # bit-exact, NaN for the zero vector included


def test_frames_bit_exact(oracle, golden_frames, T):
    params = T.default_params()
    sprite = oracle.tile_floor()
    for name, (m, aabbs, light) in golden_frames.items():
        grid = oracle.bin(params, aabbs)
        assert visible_hash(grid.count, grid.map, grid.bins) == m["grid_visible"], name
        out = oracle.render(params, aabbs, sprite, light)
        assert sha(out["gbuf"]) == m["gbuf"], name
        assert sha(out["fb"]) == m["fb"], name
        assert sha(out["brightness"]) == m["brightness"], name
        assert sha(out["lit"]) == m["lit"], name
        assert int(out["lit"].sum()) == m["lit_count"]
        assert int((out["palidx"] == T.PALIDX_BACKGROUND).sum()) == m["background"]


def test_appendix_b_default_frame(oracle, par, appendix_b, T):
    params = T.default_params()
    aabbs = graybox(par)
    st = appendix_b["stats"]
    assert len(aabbs) == st["entities"]
    sprite = oracle.tile_floor()
    light = T.make_light(480, 160, 80)  # alt:625-626
    grid = oracle.bin(params, aabbs, GridArrays(params))
    assert sha(np.frombuffer(grid.dump(), dtype=np.uint8)) == appendix_b["frame0_grid_dump"]
    out = oracle.render(params, aabbs, sprite, light)
    assert sha(out["gbuf"]) == appendix_b["frame0_gbuf"]
    fb = out["fb"].copy()
    oracle.debug_line(params, out["gbuf"], light, 0, 0, fb)  # mouse stays (0,0) in the recorded run
    assert sha(fb) == appendix_b["frame0_rgba_with_debug_line"]
    assert int(out["lit"].sum()) == st["lit_pixels"]
    assert int((out["palidx"] == T.PALIDX_BACKGROUND).sum()) == st["background_pixels"]
    g = out["gbuf"].reshape(320, 480)
    for key, exp in appendix_b["spot"].items():
        r, c = (int(v) for v in key.split(","))
        px = g[r, c]
        assert [px["normal"]["x"], px["normal"]["y"], px["normal"]["z"], px["color"]["red"], px["y"], px["z"],
                px["entity_index"]] == exp
    hit = out["palidx"] != T.PALIDX_BACKGROUND
    reds = out["gbuf"]["color"]["red"][hit]
    for level, cnt in st["palette_pixels"].items():
        assert int((reds == int(level)).sum()) == cnt
    assert len(np.unique(out["gbuf"]["entity_index"][hit])) == st["distinct_visible_entities"]
    assert len(np.unique(fb.view(np.uint32))) == st["distinct_colors"]


def test_appendix_b_scripted_frames(oracle, par, appendix_b, T):
    params = T.default_params()
    aabbs = graybox(par)
    sprite = oracle.tile_floor()
    light = T.make_light(480, 160, 80)
    keys = appendix_b["key_script"]
    for f in range(1, len(keys) + 1):
        apply_key(keys[f - 1], aabbs, light)
        if str(f) in appendix_b["script_frames"]:
            out = oracle.render(params, aabbs, sprite, light, planes=("fb", "gbuf"))
            fb = out["fb"]
            oracle.debug_line(params, out["gbuf"], light, 0, 0, fb)
            assert sha(fb) == appendix_b["script_frames"][str(f)], f


def test_rows_and_threads_agree(oracle, par, T):
    # Row blocks are independent: the row-parallel mode equals the single-thread frame byte for byte.
    params = T.default_params(512, 300, 260)  # not multiples of the bin size
    aabbs, light = par.scene_synthetic(300, 512, 300, 260, 5)
    sprite = oracle.tile_floor()
    a = oracle.render(params, aabbs, sprite, light, nthreads=1)
    b = oracle.render(params, aabbs, sprite, light, nthreads=7)
    for k in a:
        assert a[k].tobytes() == b[k].tobytes(), k
