"""The CPU oracle against the reference's OWN functions (oracle/_ref, compiled from /root/reference where it lies).
Runs only where the reference is mounted (the build container); elsewhere the committed golden vectors stand in."""
import numpy as np
import pytest

from oracle.oracle import GridArrays

pytestmark = pytest.mark.reference


def test_constants_and_sprite(oracle, reference, T):
    assert reference.consts == {"bin": 40, "width": 480, "height": 320, "length": 320, "gx": 12, "gy": 8, "gz": 8,
                                "slots": 8}
    assert oracle.tile_floor().tobytes() == reference.tile_sprite().tobytes()
    p = T.default_params()
    pal = reference.palette()
    for i in range(4):
        assert (p.palette[i].red, p.palette[i].green, p.palette[i].blue, p.palette[i].alpha) == tuple(pal[i])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_scenes_all_stages(oracle, reference, T, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 1500))
    aabbs = np.zeros(n, dtype=T.AABB)
    aabbs["px"] = rng.integers(-30, 490, n)
    aabbs["py"] = rng.integers(-30, 250, n)
    aabbs["pz"] = rng.integers(-70, 380, n)
    aabbs["ex"] = rng.integers(1, 21, n)
    aabbs["ey"] = rng.integers(0, 21, n)
    aabbs["ez"] = rng.integers(0, 21, n)
    light = T.make_light(int(rng.integers(0, 480)), int(rng.integers(0, 160)), int(rng.integers(0, 160)))
    params = reference.params()
    h = reference.scene(aabbs)
    g_ref = GridArrays(params)
    reference.bin(h, g_ref)
    g_or = oracle.bin(params, aabbs)
    assert g_ref.dump() == g_or.dump()
    gb_ref = reference.primary(h, g_ref)
    gb_or, _ = oracle.primary(params, g_or, oracle.tile_floor())
    assert gb_ref.tobytes() == gb_or.tobytes()
    fb_ref, br_ref, lit_ref = reference.shade(g_ref, gb_ref, light)
    fb_own = reference.shade_own(g_ref, gb_ref, light)  # the reference's own loop, alt:702-760
    fb_or, br_or, lit_or = oracle.shade(params, g_or, gb_or, light)
    reference.scene_free(h)
    assert fb_own.tobytes() == fb_or.tobytes()
    assert fb_own.tobytes() == fb_ref.tobytes()
    assert fb_ref.tobytes() == fb_or.tobytes()
    assert br_ref.tobytes() == br_or.tobytes()
    assert lit_ref.tobytes() == lit_or.tobytes()


def test_shadow_walk_direct(oracle, reference, T):
    rng = np.random.default_rng(11)
    n = 700
    aabbs = np.zeros(n, dtype=T.AABB)
    aabbs["px"] = rng.integers(0, 460, n)
    aabbs["py"] = rng.integers(0, 200, n)
    aabbs["pz"] = rng.integers(0, 300, n)
    aabbs["ex"] = aabbs["ey"] = aabbs["ez"] = 20
    params = reference.params()
    h = reference.scene(aabbs)
    g = GridArrays(params)
    reference.bin(h, g)
    reference.scene_free(h)
    for _ in range(3000):
        s = (int(rng.integers(0, 12)), int(rng.integers(0, 8)), int(rng.integers(0, 8)))
        e = (int(rng.integers(0, 12)), int(rng.integers(0, 8)), int(rng.integers(0, 8)))
        ray = np.zeros(1, dtype=T.RAY)
        with np.errstate(divide="ignore"):
            d = rng.integers(-3, 4, 3).astype(np.float32) / np.float32(7)
            ray["inv_x"], ray["inv_y"], ray["inv_z"] = np.float32(1) / d
        ray["ox"], ray["oy"], ray["oz"] = rng.integers(0, 480), rng.integers(0, 200), rng.integers(0, 320)
        ent = int(rng.integers(0, n))
        assert oracle.shadow(g, s, e, ent, ray) == reference.shadow(g, s, e, ent, ray)


def test_reference_own_shading_loop_on_the_golden_frames(oracle, reference, golden_frames, T):
    """alt:702-760 and alt:763-772 compiled from where they lie (ref_shade_own / ref_debug_line_own): the RGBA frame
    of every golden case equals the committed hash (which make_golden.py took from the same functions), the
    oracle's frame and the harness replay's; the debug line equals the oracle's restatement of it."""
    from helpers import sha
    params = reference.params()
    sprite = oracle.tile_floor()
    for name, (m, aabbs, light) in golden_frames.items():
        h = reference.scene(aabbs)
        g = GridArrays(params)
        reference.bin(h, g)
        gbuf = reference.primary(h, g)
        own = reference.shade_own(g, gbuf, light)
        replay, _, _ = reference.shade(g, gbuf, light)
        reference.scene_free(h)
        assert sha(own) == m["fb"], name
        assert own.tobytes() == replay.tobytes(), name
        assert own.tobytes() == oracle.render(params, aabbs, sprite, light, planes=("fb",))["fb"].tobytes(), name
        a, b = own.copy(), own.copy()
        for mx, my in [(0, 0), (240, 160), (479, 319)]:
            pick = gbuf[my * 480 + mx:my * 480 + mx + 1]
            reference.debug_line_own(pick, mx, my, light, a)
            oracle.debug_line(params, gbuf, light, mx, my, b)
            assert a.tobytes() == b.tobytes(), (name, mx, my)
