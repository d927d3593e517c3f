"""GPU parity: the HIP path through the C ABI against the CPU oracle and the committed golden vectors. Bit-exact
on every plane: integer planes (palette index, G-buffer ints, RGBA) and the fp32 brightness plane alike."""
import os

import numpy as np
import pytest

from helpers import apply_key, graybox, sha, visible_hash

pytestmark = pytest.mark.gpu

ALL = ("fb", "gbuf", "palidx", "brightness", "lit")


def assert_planes_equal(got, exp, planes, tag=""):
    for k in planes:
        g, e = got[k], exp[k]
        if g.tobytes() != e.tobytes():
            diff = np.nonzero(g.view(np.uint8).reshape(len(g), -1) != e.view(np.uint8).reshape(len(e), -1))[0]
            first = int(diff[0])
            raise AssertionError(f"{tag}: plane {k} differs at {len(np.unique(diff))} pixels; first flat index "
                                 f"{first}: gpu={g[first]} oracle={e[first]}")


@pytest.fixture(scope="module")
def sprite(par):
    return par.tile_floor()


def test_library_is_native(par):
    import ctypes
    assert par.device_count() >= 1
    assert isinstance(par.lib(), ctypes.CDLL)


def test_golden_frames(par, oracle, golden_frames, sprite, T):
    params = T.default_params()
    with par.Renderer(params) as r:
        r.set_sprites(sprite)
        for name, (m, aabbs, light) in golden_frames.items():
            r.set_entities(aabbs)
            r.set_light(light)
            out = r.render(ALL)
            count, map_, bins = r.read_grid()
            assert visible_hash(count, map_, bins) == m["grid_visible"], name
            exp = oracle.render(params, aabbs, sprite, light)
            assert_planes_equal(out, exp, ALL, name)
            assert sha(out["gbuf"]) == m["gbuf"], name
            assert sha(out["fb"]) == m["fb"], name
            assert sha(out["brightness"]) == m["brightness"], name
            assert sha(out["lit"]) == m["lit"], name


def test_default_scene_and_script(par, oracle, appendix_b, sprite, T):
    params = T.default_params()
    aabbs = graybox(par)
    light = T.make_light(480, 160, 80)
    keys = appendix_b["key_script"]
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for f in range(0, len(keys) + 1):
            if f > 0:
                apply_key(keys[f - 1], aabbs, light)
                r.update_aabbs(aabbs[0:1], 0)
                r.set_light(light)
            want = appendix_b["frame0_rgba_with_debug_line"] if f == 0 else appendix_b["script_frames"].get(str(f))
            if want is None:
                continue
            out = r.render(("fb", "gbuf"))
            if f == 0:
                assert sha(out["gbuf"]) == appendix_b["frame0_gbuf"]
                assert r.stats().bin_insertions == appendix_b["stats"]["bin_insertions"]
            fb = out["fb"]
            par.debug_line(params, out["gbuf"][0:1], 0, light, fb)  # mouse stays (0,0) in the recorded run
            assert sha(fb) == want, f


# (512, 512, 512, 64, 12345) is BASELINE config 2 and (2048, 2048, 2048, 256, 12345) config 3, both exactly as stated
@pytest.mark.parametrize("w,h,l,n,seed", [(128, 128, 128, 40, 3), (512, 512, 512, 64, 12345), (500, 333, 290, 200, 1),
                                          (1024, 640, 512, 512, 2), (2048, 2048, 2048, 256, 12345)])
def test_sizes_vs_oracle(par, oracle, sprite, T, w, h, l, n, seed):
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(n, w, h, l, seed)
    exp = oracle.render(params, aabbs, sprite, light, nthreads=os.cpu_count() or 8)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        out = r.render(ALL)
        assert_planes_equal(out, exp, ALL, f"{w}x{h}")
        # the default (skip output-neutral background shadow rays) path must give the same picture
        fast = r.render(("fb", "palidx", "brightness", "gbuf"))
        assert_planes_equal(fast, exp, ("fb", "palidx", "brightness", "gbuf"), f"{w}x{h} fast")
        st_rays = r.render(("fb",), flags=par.RENDER_COUNT_RAYS)
        assert st_rays["fb"].tobytes() == exp["fb"].tobytes()
        assert r.stats().shadow_rays == int((exp["palidx"] != T.PALIDX_BACKGROUND).sum())


def test_floor_scene_high_coverage(par, oracle, sprite, T):
    # a full floor + boxes: nearly every pixel hits, many bins wrap at 8 (alt:262-264), long shadow walks
    w, h, l = 800, 600, 600
    params = T.default_params(w, h, l)
    rows = [(i * 20, 0, j * 20, 20, 20, 20) for i in range(w // 20) for j in range(l // 20)]
    rng = np.random.default_rng(4)
    rows += [(int(rng.integers(0, w - 20)), int(rng.integers(20, 150)), int(rng.integers(0, l - 20)), 20, 20, 20)
             for _ in range(300)]
    aabbs = T.make_aabbs(rows)
    for lpos in [(500, 300, 150), (40, 20, 580), (400, 0, 0), (-200, 900, -100)]:
        light = T.make_light(*lpos)
        exp = oracle.render(params, aabbs, sprite, light, nthreads=8)
        with par.Renderer(params) as r:
            r.set_scene(aabbs, sprite, light)
            assert_planes_equal(r.render(ALL), exp, ALL, f"floor light {lpos}")


def test_row_blocks_equal_full_frame(par, oracle, sprite, T):
    w, h, l = 640, 480, 480
    params = T.default_params(w, h, l)
    aabbs, light = par.scene_synthetic(400, w, h, l, 9)
    exp = oracle.render(params, aabbs, sprite, light, nthreads=8)
    with par.Renderer(params) as r:
        r.set_scene(aabbs, sprite, light)
        for r0, r1 in [(0, 120), (120, 240), (240, 480), (37, 203), (479, 480)]:
            out = r.render(ALL, rows=(r0, r1))
            for k in ALL:
                assert out[k].tobytes() == exp[k][r0 * w:r1 * w].tobytes(), (k, r0, r1)


def test_multiple_sprites_and_ref_layout(par, oracle, T):
    w, h, l = 480, 320, 320
    params = T.default_params(w, h, l)
    base = par.tile_floor()
    sprites = np.concatenate([base, base, base])
    sprites[1]["color"] = (sprites[1]["color"] + 1) % 4
    sprites[2]["depth"] = np.maximum(sprites[2]["depth"] - 3, 0)
    sprites[2]["normal"]["x"] = 0.5
    aabbs, light = par.scene_synthetic(300, w, h, l, 21)
    ids = (np.arange(300) % 3).astype(np.int32)
    exp = oracle.render(params, aabbs, sprites, light, sprite_ids=ids)
    with par.Renderer(params) as r:
        r.set_sprites(sprites)
        r.set_entities(aabbs, ids)
        r.set_light(light)
        assert_planes_equal(r.render(ALL), exp, ALL, "sprite table")
    with par.Renderer(params) as r:  # the reference's own surface: one Sprite per entity (alt:95)
        r.set_entities_ref_layout(aabbs, sprites[ids])
        r.set_light(light)
        assert_planes_equal(r.render(ALL), exp, ALL, "ref layout")


def test_errors(par, T):
    params = T.default_params()
    with par.Renderer(params) as r:
        with pytest.raises(par.ParError) as e:
            r.render()
        assert e.value.status == 8  # NOT_READY
        r.set_sprites(par.tile_floor())
        with pytest.raises(par.ParError) as e:
            r.set_entities(T.make_aabbs([(0, 0, 0, 21, 20, 20)]))
        assert e.value.status == 6  # EXTENT
        with pytest.raises(par.ParError) as e:
            r.set_entities(T.make_aabbs([(0, 0, 0, 20, 30, 20)]))
        assert e.value.status == 6
    bad = T.default_params()
    bad.ambient = 1.5
    with pytest.raises(par.ParError):
        par.Renderer(bad)


def test_device_units_equal_the_reference(par, T):
    """AABB::intersect (alt:40-83), Color::operator* (spr:8-16) and Vector::normalize (spr:28-35) as the DEVICE
    kernels compute them (slab_hit, color_scale, normalize_l1_and_inverse through the par_debug_units hook) on the unit vectors
    tests/golden/ref_units.npz holds, against the answers the reference's own compiled functions gave for them:
    4 096 + 2 048 + 2 048 results, bit for bit (NaN payloads and +-inf inverse directions included)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_units.npz"))
    boxes = np.ascontiguousarray(z["boxes"]).view(T.AABB).reshape(-1)
    rays = np.ascontiguousarray(z["rays"]).view(T.RAY).reshape(-1)
    assert np.isnan(rays["inv_x"]).any() and np.isinf(rays["inv_y"]).any()
    hit = par.debug_units(0, boxes, rays)
    assert np.array_equal(hit, z["hit"])
    # ... and as the render kernel tests a shadow walk's records (float planes, packed arithmetic; hardware min / max
    # where the inverse direction is finite, the reference's compare-selects otherwise or throughout)
    assert np.array_equal(par.debug_units(3, boxes, rays), z["hit"])
    assert np.array_equal(par.debug_units(4, boxes, rays), z["hit"])
    cs = par.debug_units(1, z["cs_in"].astype(np.float32))
    assert np.array_equal(cs, z["cs_out"])
    nv = par.debug_units(2, z["nv"].astype(np.float32))
    assert nv.tobytes() == z["nv_out"].astype(np.float32).tobytes()


def test_short_division_sequences_are_exact(tmp_path):
    """The shading's six divisions per pixel run as a reciprocal plus two or three fused refinements
    (csrc/par_fastdiv.h) wherever the operands are in the range on which that is EXACTLY the IEEE quotient. That
    range is checked exhaustively here, on the GPU, by tools/divcheck.hip built from the same header: every a / b
    with integers |a| <= 65535, 1 <= b <= 196605, |a| <= b (2 x 10^10 quotients), and 1 / t for every float t with
    2^-24 <= |t| <= 2^24, +-0, +-inf and all NaNs, against hipcc's correctly rounded division."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "divcheck")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-Wno-unused-result",
                    "-o", exe, os.path.join(root, "tools", "divcheck.hip")], check=True, capture_output=True, timeout=300)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count(": 0 mismatches") == 2, p.stdout


def test_strip_order_visits_every_pixel_once(tmp_path):
    """A render work item visits its rectangle in vertical strips of a sprite's width (csrc/par_strips.h), with the
    divisions done through the hardware's approximate reciprocal. tools/stripcheck.hip, built from the same header,
    compares strip, column and row of EVERY pixel of EVERY rectangle shape (1..160 x 1..160: 166 million pixels)
    with plain integer arithmetic on the GPU, and that the idle lanes past a rectangle keep their strip in range."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "stripcheck")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
                    "-I", os.path.join(root, "pixel-art-raytracer_amd", "csrc"), "-o", exe,
                    os.path.join(root, "tools", "stripcheck.hip")], check=True, capture_output=True, timeout=300)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "165894400 pixels" in p.stdout and ": 0 mismatches" in p.stdout, p.stdout
