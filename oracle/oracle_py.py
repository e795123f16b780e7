"""TEST INFRASTRUCTURE — ctypes loader for oracle/libptr_oracle.so (the float64 CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libptr_oracle.so")

QUIRK_ROW_SHIFT = 1


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("primary", "hits", "shadow_rays", "exit_tca", "exit_d2", "exit_hit", "light_terms")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Hit(C.Structure):
    _fields_ = [("element", C.c_int32), ("tnear", C.c_double),
                ("point", C.c_double * 3), ("normal", C.c_double * 3)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_intersect.argtypes = [dp, dp, dp, C.c_double, dp, dp]
        L.orc_intersect.restype = C.c_int
        L.orc_trace.argtypes = [dp, C.c_int, dp, dp, dp, dp, C.POINTER(Hit), C.POINTER(Counters)]
        L.orc_trace.restype = None
        L.orc_request_pixel.argtypes = [dp, C.c_int, dp, dp, dp]
        L.orc_request_pixel.restype = None
        L.orc_trace_sample.argtypes = [dp, C.c_int, dp, dp, dp, dp, dp, dp, dp]
        L.orc_trace_sample.restype = None
        L.orc_render.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                 C.POINTER(C.c_uint32), C.POINTER(Counters)]
        L.orc_render.restype = None
        L.orc_render_mt.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                    C.POINTER(C.c_uint32), C.c_int]
        L.orc_render_mt.restype = None
        L.orc_render_rgb64.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.orc_render_rgb64.restype = None
        L.orc_trace_ext.argtypes = [dp, C.c_int, dp, dp, dp, C.c_int, dp]
        L.orc_trace_ext.restype = None
        L.orc_render_ext.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_uint32, C.POINTER(C.c_uint32), dp, C.c_int]
        L.orc_render_ext.restype = None
        L.orc_count_ext.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_uint32, C.POINTER(Counters), C.POINTER(C.c_uint64)]
        L.orc_count_ext.restype = None
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_philox4x32_10.restype = None
        ip = C.POINTER(C.c_int32)
        L.orc_rc_reproject_frame.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, ip, ip, dp]
        L.orc_rc_reproject_frame.restype = None
        up = C.POINTER(C.c_uint32)
        L.orc_rc_shade_frame.argtypes = [C.c_int, C.c_int, C.c_int, ip, dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double,
                                         dp, dp, dp, dp, up, up, up]
        L.orc_rc_shade_frame.restype = None
        L.orc_pack.argtypes = [dp]
        L.orc_pack.restype = C.c_uint32
        L.orc_reference_angle.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def scene_to_soa(elements):
    """Serialized spheres (Sphere.serialize wire format, Sphere.js:53-69) -> (12, n) float64 planes
    in the C-ABI order cx cy cz radius sr sg sb refl transp er eg eb."""
    n = len(elements)
    soa = np.zeros((12, n), dtype=np.float64)
    for i, e in enumerate(elements):
        m = e["material"]
        soa[0:3, i] = e["center"]
        soa[3, i] = e["radius"]
        soa[4:7, i] = m["surfaceColor"]
        soa[7, i] = m["reflection"]
        soa[8, i] = m["transparency"]
        soa[9:12, i] = m["emissionColor"]
    return np.ascontiguousarray(soa)


def load_scenes(golden_dir):
    with open(os.path.join(golden_dir, "scenes.json")) as f:
        return json.load(f)


def render(soa, bg, width, height, start_y=0, scan_height=None, flags=QUIRK_ROW_SHIFT,
           counters=False, threads=1):
    if scan_height is None:
        scan_height = height
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    n = soa.shape[1]
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    out = np.empty(width * scan_height, dtype=np.uint32)
    outp = out.ctypes.data_as(C.POINTER(C.c_uint32))
    if threads > 1:
        lib().orc_render_mt(_dp(soa), n, _dp(bg), width, height, start_y, scan_height, flags, outp, threads)
        return out
    cnt = Counters()
    lib().orc_render(_dp(soa), n, _dp(bg), width, height, start_y, scan_height, flags, outp,
                     C.byref(cnt) if counters else None)
    return (out, cnt.as_dict()) if counters else out


def render_rgb64(soa, bg, width, height, start_y=0, scan_height=None):
    if scan_height is None:
        scan_height = height
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    out = np.empty((scan_height, width, 3), dtype=np.float64)
    lib().orc_render_rgb64(_dp(soa), soa.shape[1], _dp(bg), width, height, start_y, scan_height, _dp(out))
    return out


def trace(soa, bg, o, d, want_hit=False):
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    rgb = np.empty(3, dtype=np.float64)
    hit = Hit()
    lib().orc_trace(_dp(soa), soa.shape[1], _dp(bg), _dp(o), _dp(d), _dp(rgb), C.byref(hit), None)
    if want_hit:
        return rgb, dict(element=hit.element, tnear=hit.tnear, point=np.array(hit.point[:]),
                         normal=np.array(hit.normal[:]))
    return rgb


def trace_batch(soa, bg, origins, dirs):
    """returns (rgb[n,3], element[n], point[n,3], normal[n,3], tnear[n])"""
    origins = np.ascontiguousarray(origins, dtype=np.float64)
    dirs = np.ascontiguousarray(dirs, dtype=np.float64)
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    n = dirs.shape[0]
    rgb = np.empty((n, 3)); el = np.empty(n, dtype=np.int32)
    pt = np.empty((n, 3)); nr = np.empty((n, 3)); tn = np.empty(n)
    hit = Hit()
    tmp = np.empty(3)
    L = lib()
    for k in range(n):
        L.orc_trace(_dp(soa), soa.shape[1], _dp(bg), _dp(origins[k]), _dp(dirs[k]), _dp(tmp), C.byref(hit), None)
        rgb[k] = tmp; el[k] = hit.element; pt[k] = hit.point[:]; nr[k] = hit.normal[:]; tn[k] = hit.tnear
    return rgb, el, pt, nr, tn


def trace_samples(soa, bg, origins, dirs):
    """render-cache samples: returns (rgb[n,3], rgb8[n,3], hit[n,3], normal[n,3]); origins (3,) or (n,3)"""
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    n = dirs.shape[0]
    origins = np.ascontiguousarray(np.broadcast_to(np.asarray(origins, dtype=np.float64).reshape(-1, 3), (n, 3)))
    out = [np.empty((n, 3)) for _ in range(4)]
    L = lib()
    for k in range(n):
        L.orc_trace_sample(_dp(soa), soa.shape[1], _dp(bg), _dp(origins[k]), _dp(dirs[k]),
                           _dp(out[0][k]), _dp(out[1][k]), _dp(out[2][k]), _dp(out[3][k]))
    return tuple(out)


def request_pixel(soa, bg, d):
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    rgb = np.empty(3, dtype=np.float64)
    lib().orc_request_pixel(_dp(soa), soa.shape[1], _dp(bg), _dp(d), _dp(rgb))
    return rgb


def intersect(o, d, c, radius):
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    t0 = C.c_double(1e8); t1 = C.c_double(1e8)
    hit = lib().orc_intersect(_dp(o), _dp(d), _dp(c), float(radius), C.byref(t0), C.byref(t1))
    return bool(hit), t0.value, t1.value


def pack(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    return int(lib().orc_pack(_dp(rgb)))


# ---- extension modes (spec only; parity unpinned by the reference) ----------------------------
class ExtExtra(C.Structure):
    _fields_ = [("planes", C.POINTER(C.c_double)), ("n_planes", C.c_int), ("phong", C.POINTER(C.c_double))]


def render_ext2(soa, bg, width, height, planes=None, phong=None, start_y=0, scan_height=None, max_depth=0, spp=1, seed=0, threads=8):
    """the extension specification with planes ((14, n_planes)) and Phong terms ((2, n + n_planes): ks, exponent)"""
    if scan_height is None:
        scan_height = height
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    pl = np.ascontiguousarray(planes, dtype=np.float64) if planes is not None else np.zeros((14, 0))
    ph = np.ascontiguousarray(phong, dtype=np.float64) if phong is not None else None
    x = ExtExtra(_dp(pl), pl.shape[1], _dp(ph) if ph is not None else None)
    out = np.empty(width * scan_height, dtype=np.uint32)
    L = lib()
    L.orc_render_ext2.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(ExtExtra), C.POINTER(C.c_double)] + [C.c_int] * 6 + \
                                 [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.c_int]
    L.orc_render_ext2.restype = None
    L.orc_render_ext2(_dp(soa), soa.shape[1], C.byref(x), _dp(bg), width, height, start_y, scan_height, max_depth, spp, seed,
                      out.ctypes.data_as(C.POINTER(C.c_uint32)), None, threads)
    return out


def render_ext(soa, bg, width, height, start_y=0, scan_height=None, max_depth=0, spp=1, seed=0,
               want_rgb64=False, threads=8):
    if scan_height is None:
        scan_height = height
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    out = np.empty(width * scan_height, dtype=np.uint32)
    rgb = np.empty((scan_height, width, 3), dtype=np.float64) if want_rgb64 else None
    lib().orc_render_ext(_dp(soa), soa.shape[1], _dp(bg), width, height, start_y, scan_height, max_depth, spp,
                         seed, out.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(rgb) if want_rgb64 else None, threads)
    return (out, rgb) if want_rgb64 else out


def count_ext(soa, bg, width, height, max_depth=0, spp=1, seed=0):
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    cnt = Counters(); sec = C.c_uint64(0)
    lib().orc_count_ext(_dp(soa), soa.shape[1], _dp(bg), width, height, 0, height, max_depth, spp, seed,
                        C.byref(cnt), C.byref(sec))
    d = cnt.as_dict(); d["secondary"] = int(sec.value)
    return d


def trace_ext(soa, bg, o, d, max_depth):
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    rgb = np.empty(3, dtype=np.float64)
    lib().orc_trace_ext(_dp(soa), soa.shape[1], _dp(bg), _dp(o), _dp(d), max_depth, _dp(rgb))
    return rgb


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


# ---- render cache: reprojection pass ------------------------------------------------------------------------
def rc_reproject(cam_from, view, proj, width, height, hits, age, age_factor=1.0, depth_threshold=1e6):
    """returns (age_out[n], item_pixel[n], pixel_element[(h+2)*(w+2)], pixel_depth[(h+2)*(w+2)]); grid borders hold
    -2 / NaN (never written, as in the reference)"""
    cam_from = np.ascontiguousarray(cam_from, dtype=np.float64); view = np.ascontiguousarray(view, dtype=np.float64)
    proj = np.ascontiguousarray(proj, dtype=np.float64); hits = np.ascontiguousarray(hits, dtype=np.float64).reshape(-1, 3)
    n = hits.shape[0]
    age = np.array(age, dtype=np.float64, copy=True)
    item_pixel = np.empty(n, dtype=np.int32)
    cells = (width + 2) * (height + 2)
    el = np.full(cells, -2, dtype=np.int32); dep = np.full(cells, np.nan)
    ip = C.POINTER(C.c_int32)
    lib().orc_rc_reproject_frame(_dp(cam_from), _dp(view), _dp(proj), width, height, n, _dp(hits), _dp(age), age_factor,
                                 depth_threshold, item_pixel.ctypes.data_as(ip), el.ctypes.data_as(ip), _dp(dep))
    return age, item_pixel, el, dep


def rc_shade(width, height, pixel_element, pixel_depth, item_age, item_rgb, age_factor=1.0, zero=20.0, rnd=5.0, pmax=255.0,
             pixel_rgb_in=None):
    """depth culling + gap filling + frames; returns dict(age, pixel_rgb, pixel_priority, hist, stats, color, reprojection, priority)"""
    n = len(item_age)
    el = np.ascontiguousarray(pixel_element, dtype=np.int32); dep = np.ascontiguousarray(pixel_depth, dtype=np.float64)
    age = np.array(item_age, dtype=np.float64, copy=True); rgb = np.ascontiguousarray(item_rgb, dtype=np.float64).reshape(-1, 3)
    cells = (width + 2) * (height + 2)
    prgb = np.zeros((cells, 3)); pprio = np.zeros(cells); hist = np.zeros(3 * 256); stats = np.zeros(3)
    if pixel_rgb_in is not None:                 # border cells that carry an element are read with the colour given here
        prgb[:] = np.asarray(pixel_rgb_in, dtype=np.float64).reshape(cells, 3)
    frames = [np.zeros(width * height, dtype=np.uint32) for _ in range(3)]
    ip = C.POINTER(C.c_int32); up = C.POINTER(C.c_uint32)
    lib().orc_rc_shade_frame(width, height, n, el.ctypes.data_as(ip), _dp(dep), _dp(age), _dp(rgb), age_factor, zero, rnd, pmax,
                             _dp(prgb), _dp(pprio), _dp(hist), _dp(stats), *[f.ctypes.data_as(up) for f in frames])
    return dict(age=age, pixel_rgb=prgb, pixel_priority=pprio, hist=hist.reshape(3, 256), stats=stats,
                color=frames[0], reprojection=frames[1], priority=frames[2])
