"""pixel-art-raytracer_amd — MI355X-native drop-in for the render call of Cons-Cat/Pixel-Art-Raytracer.

Python plumbing (ctypes) over the C ABI of libpar_raytracer.so (include/par_raytracer.h). The product is the
shared library: hand-written HIP kernels for gfx950 behind an extern-"C" boundary that replaces
src/alternative.cpp:690-760 of the reference. This module only moves pointers around; it computes nothing and has
no CPU rendering path — without the library, or without a gfx950 GPU, rendering fails loudly.

The package directory name carries a hyphen, so import it with
    par = importlib.import_module("pixel-art-raytracer_amd")
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

from . import types as T
from .types import (AABB, COLOR, LIGHT, PIXEL, SPRITE, FrameStats, Outputs, Params, default_params, make_aabbs,
                    make_light, ptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpar_raytracer.so")

PAR_OK = 0
STATUS_NAMES = {0: "PAR_OK", 1: "PAR_ERR_INVALID_ARG", 2: "PAR_ERR_NO_DEVICE", 3: "PAR_ERR_HIP", 4: "PAR_ERR_OOM",
                5: "PAR_ERR_UNSUPPORTED", 6: "PAR_ERR_EXTENT", 7: "PAR_ERR_SPRITE_ID", 8: "PAR_ERR_NOT_READY", 9: "PAR_ERR_DEVICE"}
RENDER_TRACE_BACKGROUND = 1 << 0
RENDER_COUNT_RAYS = 1 << 1
RENDER_PIPELINED = 1 << 2
RENDER_TIMED_AS_LAUNCHED = 1 << 3

# every symbol include/par_raytracer.h declares
ABI_SYMBOLS = (
    "par_status_string", "par_last_error", "par_default_params", "par_grid_dims", "par_device_count", "par_create",
    "par_destroy", "par_set_sprites", "par_set_entities", "par_set_entities_ref_layout", "par_update_aabbs",
    "par_update_aabbs_async",
    "par_set_light", "par_render", "par_render_rows", "par_render_device", "par_render_device_timed",
    "par_graph_capture", "par_graph_stage", "par_graph_launch", "par_pick", "par_get_stats", "par_read_grid",
    "par_sprite_tile_floor", "par_scene_graybox", "par_scene_synthetic", "par_debug_line", "par_debug_units",
    "par_render_device_slots", "par_row_block", "par_scene_tiles", "par_tiles_pack", "par_tiles_unpack",
    "par_background_fill", "par_tiles_assemble", "par_scene_tile_map",
)


class ParError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {detail}")


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7)
    and link it by FILE name; our library links the SONAME. If ours is loaded first the dynamic loader brings in
    /opt/rocm's copy and a later `import torch` adds a second runtime that sees no device. Loading torch's copy
    first (when torch is installed and not yet imported) makes both resolve to the same one; with torch already
    imported, or absent, there is nothing to do."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load libpar_raytracer.so (built in-tree by __graft_entry__.build() / csrc/Makefile). No fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C pixel-art-raytracer_amd/csrc` "
                              "(or __graft_entry__.build()). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.par_status_string.restype = C.c_char_p
        L.par_status_string.argtypes = [i32]
        L.par_last_error.restype = C.c_char_p
        L.par_last_error.argtypes = [vp]
        L.par_default_params.restype = None
        L.par_default_params.argtypes = [vp]
        L.par_grid_dims.argtypes = [vp, vp, vp, vp]
        L.par_device_count.argtypes = []
        L.par_create.argtypes = [vp, i32, vp]
        L.par_destroy.restype = None
        L.par_destroy.argtypes = [vp]
        L.par_set_sprites.argtypes = [vp, vp, i32]
        L.par_set_entities.argtypes = [vp, vp, vp, i32]
        L.par_set_entities_ref_layout.argtypes = [vp, vp, vp, i32]
        L.par_update_aabbs.argtypes = [vp, vp, i32, i32]
        L.par_update_aabbs_async.argtypes = [vp, vp, i32, i32, vp]
        L.par_set_light.argtypes = [vp, vp]
        L.par_render.argtypes = [vp, vp, C.c_uint]
        L.par_render_rows.argtypes = [vp, i32, i32, vp, C.c_uint]
        L.par_render_device.argtypes = [vp, vp, i32, i32, vp, C.c_uint]
        L.par_render_device_timed.argtypes = [vp, vp, i32, i32, vp, C.c_uint, vp]
        L.par_graph_capture.argtypes = [vp, vp, i32, i32, vp, C.c_uint]
        L.par_graph_stage.argtypes = [vp, vp, i32, i32, vp]
        L.par_graph_launch.argtypes = [vp, vp]
        L.par_pick.argtypes = [vp, i32, i32, vp]
        L.par_get_stats.argtypes = [vp, vp]
        L.par_read_grid.argtypes = [vp, vp, vp, vp]
        L.par_sprite_tile_floor.restype = None
        L.par_sprite_tile_floor.argtypes = [vp]
        L.par_scene_graybox.argtypes = [i32, i32, vp, i32]
        L.par_scene_synthetic.argtypes = [i32, i32, i32, i32, C.c_uint64, vp, vp]
        L.par_debug_units.argtypes = [i32, i32, vp, vp, i32, vp]
        L.par_render_device_slots.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, C.c_uint]
        L.par_row_block.restype = None
        L.par_row_block.argtypes = [i32, i32, i32, i32, vp, vp]
        L.par_scene_tiles.argtypes = [vp, vp, i32, vp, i32]
        L.par_tiles_pack.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp]
        L.par_tiles_unpack.argtypes = [vp, vp, vp, i32, vp, vp]
        L.par_background_fill.argtypes = [vp, vp, vp, i32]
        L.par_tiles_assemble.argtypes = [vp, vp, vp, vp, vp, i32, i32]
        L.par_scene_tile_map.argtypes = [vp, vp, i32, vp, i32]
        L.par_debug_read_stamps.argtypes = [vp, vp, C.c_size_t]
        L.par_debug_line.restype = None
        L.par_debug_line.argtypes = [vp, vp, i32, vp, vp]
        _lib = L
    return _lib


def debug_units(kind, in_a, in_b=None, device=0):
    """The reference's arithmetic units as the device kernels compute them (par_debug_units). kind 0: intersect
    (AABB[n], RAY[n]) -> uint8[n]; 1: color scale (float32[n, 5]) -> uint8[n, 4]; 2: normalize (float32[n, 3]) ->
    float32[n, 3]; 3, 4: intersect as the render kernel runs it on a shadow walk's records (floats, packed; 3: hardware
    min / max where the inverse direction is finite, 4: the reference's compare-selects throughout)."""
    a = np.ascontiguousarray(in_a)
    b = None if in_b is None else np.ascontiguousarray(in_b)
    n = len(a)
    out = np.zeros(n, dtype=np.uint8) if kind in (0, 3, 4) else (np.zeros((n, 4), dtype=np.uint8) if kind == 1
                                                         else np.zeros((n, 3), dtype=np.float32))
    rc = lib().par_debug_units(device, kind, ptr(a), ptr(b), n, ptr(out))
    if rc != PAR_OK:
        raise ParError(rc, "par_debug_units")
    return out


def row_block(rank, ranks, height, bin_size=40):
    """Rows [begin, end) of `rank` of `ranks` GPUs sharing one frame (par_row_block): cut at bin rows."""
    b, e = C.c_int(0), C.c_int(0)
    lib().par_row_block(rank, ranks, height, bin_size, C.byref(b), C.byref(e))
    return b.value, e.value


def device_count():
    return int(lib().par_device_count())


# ---- host-side scene helpers (C++ in the library; no GPU needed) ------------------------------------------------

def tile_floor():
    """`make_tile_floor` (spr:73-364) as a 1-element SPRITE array."""
    s = np.zeros(1, dtype=SPRITE)
    lib().par_sprite_tile_floor(ptr(s))
    return s


def scene_graybox(view_width=480, view_length=320):
    """The reference's default world, alt:517-599 (162 308 entities at 480 x 320)."""
    n = lib().par_scene_graybox(view_width, view_length, None, 0)
    a = np.zeros(n, dtype=AABB)
    lib().par_scene_graybox(view_width, view_length, ptr(a), n)
    return a


def scene_synthetic(n, width, height, length, seed):
    """SURVEY §8d synthetic scene: (aabbs, light)."""
    a = np.zeros(n, dtype=AABB)
    l = np.zeros(1, dtype=LIGHT)
    rc = lib().par_scene_synthetic(n, width, height, length, C.c_uint64(seed), ptr(a), ptr(l))
    if rc != PAR_OK:
        raise ParError(rc, "par_scene_synthetic")
    return a, l


def debug_line(params, pick_pixel, mouse_x, light, fb):
    """alt:763-772 overlay into a host frame (flat COLOR array)."""
    lib().par_debug_line(C.byref(params), ptr(pick_pixel), mouse_x, ptr(light), ptr(fb))


# ---- the renderer ------------------------------------------------------------------------------------------------

_PLANES = ("fb", "gbuf", "palidx", "brightness", "lit")
_PLANE_DTYPE = {"fb": COLOR, "gbuf": PIXEL, "palidx": np.uint8, "brightness": np.float32, "lit": np.uint8}
_PLANE_BYTES = {"fb": 4, "gbuf": 28, "palidx": 1, "brightness": 4, "lit": 1}


class Renderer:
    """One GPU's renderer: the counterpart of the reference's work arrays + render call (alt:503-517, 690-760)."""

    def __init__(self, params=None, device=-1):
        self.params = params or default_params()
        self._ctx = C.c_void_p()
        rc = lib().par_create(C.byref(self.params), device, C.byref(self._ctx))
        if rc != PAR_OK:
            raise ParError(rc, lib().par_status_string(rc).decode())
        self.width, self.height = self.params.width, self.params.height
        self._out_cache = {}

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            lib().par_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != PAR_OK:
            raise ParError(rc, lib().par_last_error(self._ctx).decode())

    # scene surface -----------------------------------------------------------------------------------------
    def set_sprites(self, sprites):
        sprites = np.ascontiguousarray(sprites, dtype=SPRITE)
        self._check(lib().par_set_sprites(self._ctx, ptr(sprites), len(sprites)))

    def set_entities(self, aabbs, sprite_ids=None):
        aabbs = np.ascontiguousarray(aabbs, dtype=AABB)
        ids = None if sprite_ids is None else np.ascontiguousarray(sprite_ids, dtype=np.int32)
        self._check(lib().par_set_entities(self._ctx, ptr(aabbs), ptr(ids), len(aabbs)))

    def set_entities_ref_layout(self, aabbs, sprite_per_entity):
        aabbs = np.ascontiguousarray(aabbs, dtype=AABB)
        sprites = np.ascontiguousarray(sprite_per_entity, dtype=SPRITE)
        assert len(aabbs) == len(sprites)
        self._check(lib().par_set_entities_ref_layout(self._ctx, ptr(aabbs), ptr(sprites), len(aabbs)))

    def update_aabbs(self, aabbs, first=0, stream=None):
        """Overwrite aabbs[first, first+len). With `stream` (the hipStream_t this renderer's frames are enqueued on)
        the copy is ordered on that stream instead of blocking."""
        aabbs = np.ascontiguousarray(aabbs, dtype=AABB)
        if stream is None:
            self._check(lib().par_update_aabbs(self._ctx, ptr(aabbs), first, len(aabbs)))
        else:
            self._check(lib().par_update_aabbs_async(self._ctx, ptr(aabbs), first, len(aabbs), C.c_void_p(stream)))

    def set_light(self, light):
        light = np.ascontiguousarray(light, dtype=LIGHT)
        self._check(lib().par_set_light(self._ctx, ptr(light)))

    def set_scene(self, aabbs, sprites, light, sprite_ids=None):
        self.set_sprites(sprites)
        self.set_entities(aabbs, sprite_ids)
        self.set_light(light)

    # render ------------------------------------------------------------------------------------------------
    def render(self, planes=("fb",), rows=None, flags=0):
        """One frame (or rows [r0, r1)) into fresh host arrays; returns {plane: flat row-major array}."""
        r0, r1 = rows or (0, self.height)
        n = (r1 - r0) * self.width
        out = {k: np.zeros(n, dtype=_PLANE_DTYPE[k]) for k in planes}
        o = Outputs(*[out[k].ctypes.data if k in out else None for k in _PLANES])
        self._check(lib().par_render_rows(self._ctx, r0, r1, C.byref(o), flags))
        return out

    def render_device(self, device_ptrs, rows=None, flags=0, stream=0, timed=False):
        """Asynchronous render into device memory. `device_ptrs` maps plane name -> raw device pointer (int) that
        addresses (row_begin, 0). `stream` is a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream)."""
        r0, r1 = rows or (0, self.height)
        # (the Outputs struct of a pointer set is cached by the pointer VALUES: this is the per-frame call of a render
        # loop, and a caller may well put a new pointer into the same dict)
        key = tuple(device_ptrs.get(k) for k in _PLANES)
        o = self._out_cache.get(key)
        if o is None:
            if len(self._out_cache) >= 64:
                self._out_cache.clear()
            o = self._out_cache[key] = Outputs(*key)
        if timed:
            st = FrameStats()
            self._check(lib().par_render_device_timed(self._ctx, C.c_void_p(stream), r0, r1, C.byref(o), flags,
                                                      C.byref(st)))
            return st
        self._check(lib().par_render_device(self._ctx, C.c_void_p(stream), r0, r1, C.byref(o), flags))
        return None

    def graph_capture(self, device_ptrs, rows=None, flags=0, stream=0):
        r0, r1 = rows or (0, self.height)
        o = Outputs(*[device_ptrs.get(k) for k in _PLANES])
        self._check(lib().par_graph_capture(self._ctx, C.c_void_p(stream), r0, r1, C.byref(o), flags))

    def graph_stage(self, aabbs=None, first=0, light=None):
        a = None if aabbs is None else np.ascontiguousarray(aabbs, dtype=AABB)
        l = None if light is None else np.ascontiguousarray(light, dtype=LIGHT)
        self._check(lib().par_graph_stage(self._ctx, ptr(a), first, 0 if a is None else len(a), ptr(l)))

    def graph_launch(self, stream=0):
        self._check(lib().par_graph_launch(self._ctx, C.c_void_p(stream)))

    def pick(self, x, y):
        px = np.zeros(1, dtype=PIXEL)
        self._check(lib().par_pick(self._ctx, x, y, ptr(px)))
        return px

    def stats(self):
        st = FrameStats()
        self._check(lib().par_get_stats(self._ctx, C.byref(st)))
        return st

    def read_grid(self):
        """(count, map, bins) of the last frame in the reference's layout (alt:503-509); parity tooling."""
        gx, gy, gz = self.params.grid_dims()
        v = gx * gy * gz
        count = np.zeros(v, dtype=np.int32)
        map_ = np.zeros(v * T.SLOTS, dtype=np.int32)
        bins = np.zeros(v * T.SLOTS, dtype=AABB)
        self._check(lib().par_read_grid(self._ctx, ptr(count), ptr(map_), ptr(bins)))
        return count, map_, bins


def plane_bytes(plane):
    return _PLANE_BYTES[plane]


def scene_tiles(params, aabbs):
    """The screen tiles (bx | by << 16, sorted by bin row then bin column) that can show a primitive of the scene
    (par_scene_tiles: host arithmetic, the same list on every rank of a sharded frame)."""
    a = np.ascontiguousarray(aabbs, dtype=AABB)
    gx, gy, _ = params.grid_dims()
    tiles = np.zeros(gx * gy, dtype=np.int32)
    n = lib().par_scene_tiles(C.byref(params), ptr(a), len(a), ptr(tiles), len(tiles))
    if n < 0:
        raise ParError(-n, "par_scene_tiles")
    return tiles[:n].copy()


def scene_tile_map(params, tiles):
    """tile column + tile row * grid-x -> index of the tile in `tiles` (its slot in a packed buffer), -1 elsewhere."""
    t = np.ascontiguousarray(tiles, dtype=np.int32)
    gx, gy, _ = params.grid_dims()
    m = np.empty(gx * gy, dtype=np.int32)
    rc = lib().par_scene_tile_map(C.byref(params), ptr(t), len(t), ptr(m), len(m))
    if rc != PAR_OK:
        raise ParError(rc, "par_scene_tile_map")
    return m


def tiles_assemble(params, d_map, packed, frame, rows, stream=0):
    """Device pointers (ints): rows [rows[0], rows[1]) of the frame at `frame` (row 0) from the packed tiles the map
    names and the background elsewhere, in one pass (asynchronous)."""
    rc = lib().par_tiles_assemble(C.byref(params), C.c_void_p(stream), C.c_void_p(d_map), C.c_void_p(packed),
                                  C.c_void_p(frame), rows[0], rows[1])
    if rc != PAR_OK:
        raise ParError(rc, "par_tiles_assemble")


def tiles_pack(params, d_tiles, n, fb_block, rows, packed, stream=0):
    """Device pointers (ints): tiles d_tiles[0, n) of the frame block `rows` at fb_block -> packed slots (asynchronous)."""
    rc = lib().par_tiles_pack(C.byref(params), C.c_void_p(stream), C.c_void_p(d_tiles), n, C.c_void_p(fb_block), rows[0],
                              rows[1], C.c_void_p(packed))
    if rc != PAR_OK:
        raise ParError(rc, "par_tiles_pack")


def tiles_unpack(params, d_tiles, n, packed, frame, stream=0):
    rc = lib().par_tiles_unpack(C.byref(params), C.c_void_p(stream), C.c_void_p(d_tiles), n, C.c_void_p(packed),
                                C.c_void_p(frame))
    if rc != PAR_OK:
        raise ParError(rc, "par_tiles_unpack")


def background_fill(params, rows_ptr, n_rows, stream=0):
    rc = lib().par_background_fill(C.byref(params), C.c_void_p(stream), C.c_void_p(rows_ptr), n_rows)
    if rc != PAR_OK:
        raise ParError(rc, "par_background_fill")
