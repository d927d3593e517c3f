"""ctypes / numpy mirrors of include/par_types.h.

Layouts follow the reference byte for byte (spr = src/sprites.hpp, alt = src/alternative.cpp):
Color spr:5-17, Vector<float> spr:20-51, Pixel spr:53-58, Sprite spr:67-71, AABB alt:35-38,88, Light alt:619-622,
Ray alt:30-33.
"""
import ctypes as C

import numpy as np

SPRITE_W = 20
SPRITE_H = 40
SPRITE_TEXELS = SPRITE_W * SPRITE_H
SLOTS = 8
MAX_PALETTE = 256
PALIDX_BACKGROUND = 0xFF

COLOR = np.dtype([("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")])
VEC3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
PIXEL = np.dtype([("normal", VEC3), ("color", COLOR), ("y", "<i4"), ("z", "<i4"), ("entity_index", "<i4")])
SPRITE = np.dtype([("color", "<i4", (SPRITE_TEXELS,)), ("depth", "<i4", (SPRITE_TEXELS,)),
                   ("normal", VEC3, (SPRITE_TEXELS,))])
AABB = np.dtype([("px", "<i2"), ("py", "<i2"), ("pz", "<i2"), ("ex", "<i2"), ("ey", "<i2"), ("ez", "<i2"),
                 ("pad", "<i2", (2,))])
LIGHT = np.dtype([("x", "<i2"), ("y", "<i2"), ("z", "<i2"), ("radius", "<i2")])
RAY = np.dtype([("inv_x", "<f4"), ("inv_y", "<f4"), ("inv_z", "<f4"), ("ox", "<i2"), ("oy", "<i2"), ("oz", "<i2"),
                ("pad", "<i2")])

assert COLOR.itemsize == 4 and VEC3.itemsize == 12 and PIXEL.itemsize == 28
assert SPRITE.itemsize == 16000 and AABB.itemsize == 16 and LIGHT.itemsize == 8 and RAY.itemsize == 20


class Color(C.Structure):
    _fields_ = [("red", C.c_uint8), ("green", C.c_uint8), ("blue", C.c_uint8), ("alpha", C.c_uint8)]


class Params(C.Structure):
    """par_params: the reference's constexpr view/grid constants (alt:116-131) as run-time values."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("length", C.c_int32), ("bin_size", C.c_int32),
                ("ambient", C.c_float), ("background", C.c_uint8), ("reserved_", C.c_uint8 * 3),
                ("palette_size", C.c_int32), ("palette", Color * MAX_PALETTE)]

    def grid_dims(self):
        b = self.bin_size
        return ((self.width + b - 1) // b, (self.height + b - 1) // b, (self.length + b - 1) // b)


class Outputs(C.Structure):
    """par_outputs: nullable output planes, each addressing the element of (row_begin, 0)."""
    _fields_ = [("fb", C.c_void_p), ("gbuf", C.c_void_p), ("palidx", C.c_void_p), ("brightness", C.c_void_p),
                ("lit", C.c_void_p)]


class FrameStats(C.Structure):
    _fields_ = [("entities", C.c_int64), ("bin_insertions", C.c_int64), ("shadow_rays", C.c_int64),
                ("occupied_columns", C.c_int64), ("overflow_columns", C.c_int64),
                ("ms_bin", C.c_float), ("ms_fill", C.c_float), ("ms_render", C.c_float),
                ("ms_overflow", C.c_float), ("ms_launch", C.c_float * 5), ("render_merged", C.c_int32)]


def default_params(width=480, height=320, length=None, bin_size=40):
    """Reference defaults (alt:116-119, 281, 702; spr:60-65) with an optional view size; length defaults to height
    as in the reference (alt:118-119)."""
    p = Params()
    p.width, p.height = width, height
    p.length = height if length is None else length
    p.bin_size = bin_size
    p.ambient = 0.25
    p.background = 255 // 2
    p.palette_size = 4
    for i, g in enumerate((100, 140, 200, 240)):
        p.palette[i] = Color(g, g, g, 0)
    return p


def make_light(x, y, z, radius=10):
    l = np.zeros(1, dtype=LIGHT)
    l["x"], l["y"], l["z"], l["radius"] = x, y, z, radius
    return l


def make_aabbs(rows):
    """AABB array from an iterable of (px, py, pz, ex, ey, ez)."""
    rows = list(rows)
    a = np.zeros(len(rows), dtype=AABB)
    for i, r in enumerate(rows):
        a[i]["px"], a[i]["py"], a[i]["pz"], a[i]["ex"], a[i]["ey"], a[i]["ez"] = r
    return a


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
