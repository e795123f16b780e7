"""ptr_amd — thin ctypes binding over the C ABI of include/ptr.h (libptr_hip.so).

Plumbing for pytest and bench.py only; the product's host language is JavaScript
(path-tracer_amd/js).  There is no CPU fallback: importing works without a GPU (so the
symbol table can be checked), creating a Renderer does not.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libptr_hip.so"))

QUIRK_ROW_SHIFT = 0x0001
MODE_EXACT_F64 = 0x0002
MODE_F32_ONLY = 0x0004
MODE_TOLERANT = 0x0020
FLAG_STATS = 0x0008
FLAG_TIMING = 0x0010
DEBUG_REASONS = 0x8000
DEBUG_FORCE_MANY = 0x4000
DEBUG_EXT_LANEWISE = 0x1000

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NO_SCENE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5

# every symbol include/ptr.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "ptr_abi_version", "ptr_create", "ptr_destroy", "ptr_last_error",
    "ptr_set_scene", "ptr_set_scene_f64", "ptr_set_scene_sequence_f64",
    "ptr_set_background", "ptr_set_background_f64", "ptr_set_camera", "ptr_set_extension_f64",
    "ptr_render", "ptr_render_device", "ptr_render_device_noalloc", "ptr_render_strips_device", "ptr_strip_rows", "ptr_strip_image_row", "ptr_sync",
    "ptr_trace_batch", "ptr_trace_batch_f64", "ptr_trace_samples_f64", "ptr_rc_reproject", "ptr_rc_reproject_device", "ptr_rc_shade", "ptr_rc_shade_device", "ptr_get_stats", "ptr_get_kernel_ms",
    "ptr_gather_plan", "ptr_comm_unique_id", "ptr_comm_init_rank", "ptr_comm_init_all", "ptr_comm_destroy",
    "ptr_gather_strips_device", "ptr_gather_strips_all", "ptr_place_strips_device", "ptr_render_multi",
    "ptr_rc_cache_create", "ptr_rc_cache_destroy", "ptr_rc_cache_shape", "ptr_rc_cache_upload", "ptr_rc_cache_download",
    "ptr_rc_next_frame", "ptr_rc_get_frames", "ptr_rc_pick_sequence",
    "ptr_rc_reproject_shard_device", "ptr_rc_merge_shards_device", "ptr_rc_reproject_multi_device",
]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("start_y", C.c_int32),
                ("scan_height", C.c_int32), ("max_depth", C.c_int32), ("spp", C.c_int32),
                ("seed", C.c_uint32), ("flags", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("right", C.c_double * 3), ("up", C.c_double * 3),
                ("forward", C.c_double * 3), ("tan_half_fov", C.c_double)]


class RcCamera(C.Structure):
    _fields_ = [("from_", C.c_double * 3), ("view", C.c_double * 16), ("proj", C.c_double * 16),
                ("width", C.c_int32), ("height", C.c_int32)]


class RcShadeParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("age_factor", C.c_double), ("interpolation_zero", C.c_double),
                ("interpolation_random", C.c_double), ("priority_max", C.c_double)]


class RcView(C.Structure):
    _fields_ = [("from_", C.c_double * 3), ("world", C.c_double * 16), ("world_inv", C.c_double * 16),
                ("proj", C.c_double * 16), ("proj_inv", C.c_double * 16)]


class RcLoopParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("age_factor", "depth_threshold", "interpolation_zero", "interpolation_random",
                                          "priority_max", "max_samples_per_frame")]


class RcFrameStats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("threshold", "request_factor", "candidates", "requests", "completeness",
                                          "total_priority", "total_priority_increment", "cache_usage")]


# the render cache's state as ptr_rc_cache_upload / _download order it: (name, dtype, entries per item or per grid cell)
RC_STATE = (("item_pixel", np.int32, "n"), ("item_hit", np.float64, "3n"), ("item_age", np.float64, "n"), ("item_rgb", np.uint32, "n"),
            ("pixel_element", np.int32, "c"), ("pixel_depth", np.float64, "c"), ("pixel_rgb", np.uint32, "c"),
            ("pixel_priority", np.int32, "c"), ("pixel_sample", np.uint8, "c"))


class Strips(C.Structure):
    _fields_ = [("n_parts", C.c_int32), ("part", C.c_int32), ("strip_height", C.c_int32),
                ("reserved", C.c_int32)]


class GatherSizes(C.Structure):
    _fields_ = [("rows_max", C.c_int32), ("rows", C.c_int32), ("frame_stride_px", C.c_uint64), ("count", C.c_uint64),
                ("local_bytes", C.c_uint64), ("root_stage_bytes", C.c_uint64), ("full_bytes", C.c_uint64),
                ("last_full_index", C.c_uint64)]


def gather_plan(width, height, strip_height, n_frames, part, n_parts, frame_stride_px=0):
    """ptr_gather_plan: the sizes of one strip gather (host arithmetic only; works without a GPU)."""
    g = GatherSizes()
    L = lib()
    L.ptr_gather_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(GatherSizes)]
    rc = L.ptr_gather_plan(width, height, strip_height, n_frames, frame_stride_px, part, n_parts, C.byref(g))
    if rc != 0:
        raise PtrError(rc, L.ptr_last_error(None).decode())
    return {k: int(getattr(g, k)) for k, _ in GatherSizes._fields_}


class StripPart:
    """Cyclic-strip share of one rank: strip k of `strip_height` rows belongs to part k % n_parts."""

    def __init__(self, n_parts, part, strip_height):
        self.c = Strips(n_parts, part, strip_height, 0)

    def local_rows(self, height):
        return lib().ptr_strip_rows(C.byref(self.c), height)

    def image_row(self, local_row):
        return lib().ptr_strip_image_row(C.byref(self.c), local_row)

    def global_rows(self, height):
        sh, n, part = self.c.strip_height, self.c.n_parts, self.c.part
        rows = []
        for k in range(part, (height + sh - 1) // sh, n):
            rows.extend(range(k * sh, min(k * sh + sh, height)))
        return rows


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("primary", "hits", "shadow_rays", "exit_tca", "exit_d2", "exit_hit", "retraced", "reshaded", "secondary")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PtrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptr error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libptr_hip.so; raises loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C path-tracer_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7.  If torch
        # is going to be used in this process (bench.py, tests that hand us torch device memory) it
        # must be loaded first so that libptr_hip.so binds to the same runtime instance; two
        # instances cannot both own the GPU ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, dp, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.ptr_abi_version.restype = C.c_int
        L.ptr_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ptr_destroy.argtypes = [vp]
        L.ptr_destroy.restype = None
        L.ptr_last_error.argtypes = [vp]
        L.ptr_last_error.restype = C.c_char_p
        L.ptr_set_scene.argtypes = [vp, fp, C.c_int]
        L.ptr_set_scene_f64.argtypes = [vp, dp, C.c_int]
        L.ptr_set_scene_sequence_f64.argtypes = [vp, dp, C.c_int, C.c_int]
        L.ptr_set_background.argtypes = [vp, fp]
        L.ptr_set_background_f64.argtypes = [vp, dp]
        L.ptr_set_camera.argtypes = [vp, C.POINTER(Camera)]
        L.ptr_set_extension_f64.argtypes = [vp, dp, C.c_int, dp, C.c_int]
        L.ptr_render.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_uint32), fp]
        L.ptr_render_device.argtypes = [vp, C.POINTER(Params), C.c_int, C.c_int, vp, vp, vp]
        L.ptr_render_device_noalloc.argtypes = [vp, C.POINTER(Params), C.c_int, C.c_int, vp, vp, vp]
        L.ptr_render_strips_device.argtypes = [vp, C.POINTER(Params), C.POINTER(Strips), C.c_int, C.c_int,
                                               vp, vp, C.c_uint64, vp]
        L.ptr_strip_rows.argtypes = [C.POINTER(Strips), C.c_int]
        L.ptr_strip_image_row.argtypes = [C.POINTER(Strips), C.c_int]
        L.ptr_sync.argtypes = [vp]
        L.ptr_trace_batch.argtypes = [vp, fp, C.c_int, C.POINTER(C.c_uint8), fp, fp]
        L.ptr_trace_batch_f64.argtypes = [vp, dp, dp, C.c_int, dp, C.POINTER(C.c_int32), dp, dp]
        L.ptr_trace_samples_f64.argtypes = [vp, dp, dp, C.c_int, dp, C.POINTER(C.c_uint8), dp, dp]
        ip = C.POINTER(C.c_int32)
        L.ptr_rc_reproject.argtypes = [vp, C.POINTER(RcCamera), C.c_int, dp, dp, C.c_double, C.c_double, ip, ip, dp]
        L.ptr_rc_reproject_device.argtypes = [vp, C.POINTER(RcCamera), C.c_int, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp]
        up = C.POINTER(C.c_uint32)
        L.ptr_rc_shade.argtypes = [vp, C.POINTER(RcShadeParams), C.c_int, ip, dp, dp, up, up, ip, up, dp, up, up, up]
        L.ptr_rc_shade_device.argtypes = [vp, C.POINTER(RcShadeParams), C.c_int] + [vp] * 12
        L.ptr_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
        L.ptr_comm_init_rank.argtypes = [vp, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
        L.ptr_comm_init_all.argtypes = [C.POINTER(vp), C.c_int]
        L.ptr_comm_destroy.argtypes = [vp]
        L.ptr_gather_strips_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_uint64, vp, C.c_int, vp]
        L.ptr_gather_strips_all.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.c_uint64, vp]
        L.ptr_place_strips_device.argtypes = [vp, C.POINTER(Strips), C.c_int, C.c_int, C.c_int, vp, C.c_uint64, vp, vp]
        L.ptr_render_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Params), C.POINTER(C.c_uint32)]
        L.ptr_rc_reproject_shard_device.argtypes = [vp, C.POINTER(RcCamera), C.c_int, C.c_int, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp]
        L.ptr_rc_merge_shards_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_double, C.c_double,
                                                 vp, vp, vp, vp]
        L.ptr_rc_reproject_multi_device.argtypes = [vp, C.POINTER(RcCamera), C.c_int, C.c_int, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp]
        L.ptr_rc_cache_create.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.ptr_rc_cache_destroy.argtypes = [vp]
        L.ptr_rc_cache_destroy.restype = None
        L.ptr_rc_cache_shape.argtypes = [vp, ip, ip, ip]
        L.ptr_rc_cache_upload.argtypes = [vp] * 10 + [C.c_int64]
        L.ptr_rc_cache_download.argtypes = [vp] * 10 + [C.POINTER(C.c_int64)]
        L.ptr_rc_next_frame.argtypes = [vp, C.POINTER(RcView), C.POINTER(RcLoopParams), C.POINTER(RcFrameStats)]
        L.ptr_rc_get_frames.argtypes = [vp, up, up, up, up]
        L.ptr_rc_pick_sequence.argtypes = [C.c_double, C.c_int, ip, C.c_int]
        L.ptr_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.ptr_get_kernel_ms.argtypes = [vp, fp, fp]
        _lib = L
    return _lib


def scene_to_soa(elements):
    """Serialized spheres (the reference's Sphere.serialize wire format, RT/src/Sphere.js:53-69)
    -> (12, n) float64 planes: cx cy cz radius sr sg sb reflection transparency er eg eb."""
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


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def rc_pick_sequence(request_factor, n_candidates):
    """the candidate numbers Driver.directSamples requests, by the library's closed form (host only)"""
    out = np.empty(max(n_candidates, 1), dtype=np.int32)
    n = lib().ptr_rc_pick_sequence(float(request_factor), int(n_candidates), out.ctypes.data_as(C.POINTER(C.c_int32)), out.size)
    if n < 0:
        raise PtrError(n, "ptr_rc_pick_sequence refused the arguments")
    return out[:n].copy()


def render_multi(renderers, width, height, flags=0, max_depth=0, spp=1, seed=0):
    """ptr_render_multi over Renderer objects that share a communicator (comm_init_all): one frame, host array out"""
    L = lib()
    n = len(renderers)
    hs = (C.c_void_p * n)(*[r._h for r in renderers])
    p = Params(width, height, 0, height, max_depth, spp, seed, flags)
    out = np.empty(width * height, dtype=np.uint32)
    rc = L.ptr_render_multi(hs, n, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != OK:
        raise PtrError(rc, L.ptr_last_error(renderers[0]._h).decode())
    return out


def comm_init_all(renderers):
    L = lib()
    n = len(renderers)
    hs = (C.c_void_p * n)(*[r._h for r in renderers])
    rc = L.ptr_comm_init_all(hs, n)
    if rc != OK:
        raise PtrError(rc, L.ptr_last_error(renderers[0]._h).decode())


class Renderer:
    """One ptr_ctx.  Mirrors `new RayTracer(backgroundColor, scene)` + `.render(...)`."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.ptr_create(device, C.byref(h))
        if rc != OK:
            raise PtrError(rc, self._L.ptr_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.ptr_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != OK:
            raise PtrError(rc, self._L.ptr_last_error(self._h).decode())

    # ---- scene
    def set_scene(self, soa):
        soa = np.ascontiguousarray(soa, dtype=np.float64)
        if soa.ndim == 2:
            assert soa.shape[0] == 12
            self._check(self._L.ptr_set_scene_f64(self._h, _dp(soa), soa.shape[1]))
        else:
            assert soa.ndim == 3 and soa.shape[1] == 12     # (frames, 12, n)
            self._check(self._L.ptr_set_scene_sequence_f64(self._h, _dp(soa), soa.shape[2], soa.shape[0]))

    def set_scene_f32(self, soa):
        soa = np.ascontiguousarray(soa, dtype=np.float32)
        self._check(self._L.ptr_set_scene(self._h, soa.ctypes.data_as(C.POINTER(C.c_float)), soa.shape[1]))

    def set_elements(self, elements):
        self.set_scene(scene_to_soa(elements))

    def set_background(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.float64)
        self._check(self._L.ptr_set_background_f64(self._h, _dp(rgb)))

    def set_extension(self, planes=None, phong=None):
        """planes: (14, n_planes) rows px py pz nx ny nz sr sg sb refl transp er eg eb; phong: (2, n_elements) rows ks, exponent"""
        pl = None if planes is None else np.ascontiguousarray(planes, dtype=np.float64)
        ph = None if phong is None else np.ascontiguousarray(phong, dtype=np.float64)
        self._check(self._L.ptr_set_extension_f64(self._h, _dp(pl) if pl is not None else None, 0 if pl is None else pl.shape[1],
                                                  _dp(ph) if ph is not None else None, 0 if ph is None else ph.shape[1]))

    def set_camera(self, origin=None, right=(1, 0, 0), up=(0, 1, 0), forward=(0, 0, -1), tan_half_fov=None):
        if origin is None:
            self._check(self._L.ptr_set_camera(self._h, None))
            return
        import math
        cam = Camera()
        cam.origin[:] = origin
        cam.right[:] = right
        cam.up[:] = up
        cam.forward[:] = forward
        cam.tan_half_fov = tan_half_fov if tan_half_fov is not None else math.tan(math.pi * 0.5 * 30 / 180)
        self._check(self._L.ptr_set_camera(self._h, C.byref(cam)))

    # ---- render
    @staticmethod
    def params(width, height, start_y=0, scan_height=None, flags=0, max_depth=0, spp=1, seed=0):
        return Params(width, height, start_y, height if scan_height is None else scan_height,
                      max_depth, spp, seed, flags)

    def render(self, width, height, start_y=0, scan_height=None, flags=QUIRK_ROW_SHIFT, want_f32=False,
               max_depth=0, spp=1, seed=0):
        p = self.params(width, height, start_y, scan_height, flags, max_depth, spp, seed)
        out = np.empty(p.width * p.scan_height, dtype=np.uint32)
        f32 = np.empty((p.scan_height, p.width, 4), dtype=np.float32) if want_f32 else None
        self._check(self._L.ptr_render(self._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       f32.ctypes.data_as(C.POINTER(C.c_float)) if want_f32 else None))
        return (out, f32) if want_f32 else out

    def render_device(self, p, first_frame, n_frames, d_rgba8, d_rgba32f=None, stream=None):
        """d_rgba8 / d_rgba32f: device addresses (ints, e.g. tensor.data_ptr()); enqueue only.
        stream: a HIP stream handle; None or 0 = the context's own (non-blocking) stream, which torch's default
        stream does not wait for — pass torch.cuda.Stream().cuda_stream of the stream the consumers run on."""
        self._check(self._L.ptr_render_device(self._h, C.byref(p), first_frame, n_frames,
                                              C.c_void_p(d_rgba8),
                                              C.c_void_p(d_rgba32f) if d_rgba32f else None,
                                              C.c_void_p(stream) if stream else None))

    def render_device_noalloc(self, p, first_frame, n_frames, d_rgba8, d_rgba32f=None, stream=None):
        """ptr_render_device_noalloc: same outputs as render_device, allocates nothing (scene must be resident: sync() first)."""
        self._check(self._L.ptr_render_device_noalloc(self._h, C.byref(p), first_frame, n_frames,
                                                      C.c_void_p(d_rgba8),
                                                      C.c_void_p(d_rgba32f) if d_rgba32f else None,
                                                      C.c_void_p(stream) if stream else None))

    def render_strips_device(self, p, part, first_frame, n_frames, d_rgba8, d_rgba32f=None, stream=None,
                             frame_stride_px=0):
        self._check(self._L.ptr_render_strips_device(
            self._h, C.byref(p), C.byref(part.c), first_frame, n_frames, C.c_void_p(d_rgba8),
            C.c_void_p(d_rgba32f) if d_rgba32f else None, frame_stride_px,
            C.c_void_p(stream) if stream else None))

    def sync(self):
        self._check(self._L.ptr_sync(self._h))

    # ---- single rays
    def trace_batch_f64(self, dirs, origins=None, want_hit=False):
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = dirs.shape[0]
        if origins is not None:
            origins = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
            assert origins.shape[0] == n
        rgb = np.empty((n, 3), dtype=np.float64)
        el = np.empty(n, dtype=np.int32)
        hit = np.empty((n, 3), dtype=np.float64)
        nrm = np.empty((n, 3), dtype=np.float64)
        self._check(self._L.ptr_trace_batch_f64(
            self._h, _dp(origins) if origins is not None else None, _dp(dirs), n, _dp(rgb),
            el.ctypes.data_as(C.POINTER(C.c_int32)), _dp(hit), _dp(nrm)))
        return (rgb, el, hit, nrm) if want_hit else rgb

    def trace_samples(self, origin, dirs):
        """render-cache samples: returns (rgb64[n,3], rgb8[n,3], hit[n,3], normal[n,3])"""
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        origin = np.ascontiguousarray(origin, dtype=np.float64).reshape(3)
        n = dirs.shape[0]
        rgb = np.empty((n, 3)); rgb8 = np.empty((n, 3), dtype=np.uint8); hit = np.empty((n, 3)); nrm = np.empty((n, 3))
        self._check(self._L.ptr_trace_samples_f64(self._h, _dp(origin), _dp(dirs), n, _dp(rgb),
                                                  rgb8.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(hit), _dp(nrm)))
        return rgb, rgb8, hit, nrm

    @staticmethod
    def rc_camera(cam_from, view, proj, width, height):
        cam = RcCamera()
        cam.from_[:] = [float(v) for v in cam_from]
        cam.view[:] = [float(v) for v in view]
        cam.proj[:] = [float(v) for v in proj]
        cam.width, cam.height = width, height
        return cam

    def rc_reproject(self, cam_from, view, proj, width, height, hits, age, age_factor=1.0, depth_threshold=1e6):
        """render cache reprojection pass on host arrays; returns (age_out, item_pixel, pixel_element, pixel_depth)
        with the never-written grid border holding -2 / NaN"""
        cam = self.rc_camera(cam_from, view, proj, width, height)
        hits = np.ascontiguousarray(hits, dtype=np.float64).reshape(-1, 3)
        n = hits.shape[0]
        age = np.array(age, dtype=np.float64, copy=True)
        item_pixel = np.empty(n, dtype=np.int32)
        cells = (width + 2) * (height + 2)
        el = np.full(cells, -2, dtype=np.int32); dep = np.full(cells, np.nan)
        ip = C.POINTER(C.c_int32)
        self._check(self._L.ptr_rc_reproject(self._h, C.byref(cam), n, _dp(hits), _dp(age), age_factor, depth_threshold,
                                             item_pixel.ctypes.data_as(ip), el.ctypes.data_as(ip), _dp(dep)))
        return age, item_pixel, el, dep

    def rc_shade(self, width, height, pixel_element, pixel_depth, item_age, item_rgb, age_factor=1.0, zero=20.0, rnd=5.0, pmax=255.0,
                 pixel_rgb_in=None):
        """depth culling + gap filling + frames on host arrays (item_rgb: (n, 3) byte values); same dictionary keys as the tests' CPU checker.
        pixel_rgb_in: (cells, 3) byte values, the grid's colours before the pass (read for border cells that carry an element)"""
        p = RcShadeParams(width, height, age_factor, zero, rnd, pmax)
        el = np.ascontiguousarray(pixel_element, dtype=np.int32); dep = np.ascontiguousarray(pixel_depth, dtype=np.float64)
        age = np.array(item_age, dtype=np.float64, copy=True)
        c = np.asarray(item_rgb, dtype=np.float64).reshape(-1, 3).astype(np.uint32)
        rgb = np.ascontiguousarray(c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16), dtype=np.uint32)
        cells = (width + 2) * (height + 2)
        prgb = np.zeros(cells, dtype=np.uint32); pprio = np.zeros(cells, dtype=np.int32)
        if pixel_rgb_in is not None:
            q = np.asarray(pixel_rgb_in, dtype=np.float64).reshape(cells, 3).astype(np.uint32)
            prgb[:] = q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16)
        hist = np.zeros(768, dtype=np.uint32); stats = np.zeros(3)
        frames = [np.zeros(width * height, dtype=np.uint32) for _ in range(3)]
        ip = C.POINTER(C.c_int32); up = C.POINTER(C.c_uint32)
        self._check(self._L.ptr_rc_shade(self._h, C.byref(p), len(age), el.ctypes.data_as(ip), _dp(dep), _dp(age), rgb.ctypes.data_as(up),
                                         prgb.ctypes.data_as(up), pprio.ctypes.data_as(ip), hist.ctypes.data_as(up), _dp(stats),
                                         *[f.ctypes.data_as(up) for f in frames]))
        pixel_rgb = np.stack([prgb & 255, (prgb >> 8) & 255, (prgb >> 16) & 255], axis=1).astype(np.float64)
        return dict(age=age, pixel_rgb=pixel_rgb, pixel_priority=pprio.astype(np.float64), hist=hist.reshape(3, 256).astype(np.float64),
                    stats=stats, color=frames[0], reprojection=frames[1], priority=frames[2])

    def rc_reproject_device(self, cam, n, d_hit, d_age, d_item_pixel, d_pel, d_pdep, age_factor=1.0, depth_threshold=1e6, stream=None):
        self._check(self._L.ptr_rc_reproject_device(self._h, C.byref(cam), n, C.c_void_p(d_hit), C.c_void_p(d_age), age_factor,
                                                    depth_threshold, C.c_void_p(d_item_pixel), C.c_void_p(d_pel),
                                                    C.c_void_p(d_pdep), C.c_void_p(stream) if stream else None))

    # ---- the render cache's frame loop on a cache resident on the GPU (ptr_rc_cache_*, ptr_rc_next_frame) ----
    def rc_cache_create(self, width, height, n_items):
        self._check(self._L.ptr_rc_cache_create(self._h, width, height, n_items))
        self._rc_shape = (width, height, n_items)

    def rc_cache_destroy(self):
        self._L.ptr_rc_cache_destroy(self._h)

    def _rc_sizes(self):
        w, h, n = self._rc_shape
        return {"n": n, "3n": 3 * n, "c": (w + 2) * (h + 2)}

    def rc_cache_upload(self, state, cache_pointer):
        """state: dict of the arrays RC_STATE names (numpy, converted to the dtypes listed there)"""
        sizes = self._rc_sizes()
        arrs = []
        for name, dt, size in RC_STATE:
            a = np.ascontiguousarray(state[name], dtype=dt).reshape(-1)
            if a.size != sizes[size]:
                raise ValueError(f"{name} needs {sizes[size]} entries, has {a.size}")
            arrs.append(a)
        self._check(self._L.ptr_rc_cache_upload(self._h, *[C.c_void_p(a.ctypes.data) for a in arrs], int(cache_pointer)))

    def rc_cache_download(self):
        sizes = self._rc_sizes()
        out = {name: np.empty(sizes[size], dtype=dt) for name, dt, size in RC_STATE}
        cp = C.c_int64(0)
        self._check(self._L.ptr_rc_cache_download(self._h, *[C.c_void_p(out[name].ctypes.data) for name, _, _ in RC_STATE], C.byref(cp)))
        return out, cp.value

    def rc_next_frame(self, cam_from, world, world_inv, proj, proj_inv, max_samples_per_frame, age_factor=1.0, depth_threshold=1e6,
                      zero=20.0, rnd=5.0, pmax=255.0):
        v = RcView()
        v.from_[:] = [float(x) for x in cam_from]
        for name, m in (("world", world), ("world_inv", world_inv), ("proj", proj), ("proj_inv", proj_inv)):
            getattr(v, name)[:] = [float(x) for x in np.asarray(m, dtype=np.float64).reshape(16)]
        p = RcLoopParams(age_factor, depth_threshold, zero, rnd, pmax, max_samples_per_frame)
        st = RcFrameStats()
        self._check(self._L.ptr_rc_next_frame(self._h, C.byref(v), C.byref(p), C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def rc_get_frames(self, which=("color", "reprojection", "priority", "sampling")):
        w, h, _ = self._rc_shape
        names = ("color", "reprojection", "priority", "sampling")
        out = {k: np.empty(w * h, dtype=np.uint32) for k in which}
        ptrs = [out[k].ctypes.data_as(C.POINTER(C.c_uint32)) if k in out else None for k in names]
        self._check(self._L.ptr_rc_get_frames(self._h, *ptrs))
        return out

    # ---- reprojection of a cache split over several GPUs (device pointers as integers) ----
    def rc_reproject_shard_device(self, cam, n_local, item_base, d_hit, d_age, d_item_pixel, d_local_el, d_local_dep,
                                  age_factor=1.0, depth_threshold=1e6, stream=None):
        self._check(self._L.ptr_rc_reproject_shard_device(self._h, C.byref(cam), n_local, item_base, C.c_void_p(d_hit), C.c_void_p(d_age),
                                                          age_factor, depth_threshold, C.c_void_p(d_item_pixel), C.c_void_p(d_local_el),
                                                          C.c_void_p(d_local_dep), C.c_void_p(stream or 0)))

    def rc_merge_shards_device(self, width, height, n_ranks, rank, d_all_el, d_all_dep, n_local, item_base, d_age, d_item_pixel, d_pel,
                               d_pdep, age_factor=1.0, depth_threshold=1e6, stream=None):
        self._check(self._L.ptr_rc_merge_shards_device(self._h, width, height, n_ranks, rank, C.c_void_p(d_all_el), C.c_void_p(d_all_dep),
                                                       n_local, item_base, C.c_void_p(d_age), age_factor, depth_threshold,
                                                       C.c_void_p(d_item_pixel), C.c_void_p(d_pel), C.c_void_p(d_pdep), C.c_void_p(stream or 0)))

    def rc_reproject_multi_device(self, cam, n_local, item_base, d_hit, d_age, d_item_pixel, d_pel, d_pdep, age_factor=1.0,
                                  depth_threshold=1e6, stream=None):
        self._check(self._L.ptr_rc_reproject_multi_device(self._h, C.byref(cam), n_local, item_base, C.c_void_p(d_hit), C.c_void_p(d_age),
                                                          age_factor, depth_threshold, C.c_void_p(d_item_pixel), C.c_void_p(d_pel),
                                                          C.c_void_p(d_pdep), C.c_void_p(stream or 0)))

    def trace_batch(self, dirs, want_hit=False):
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = dirs.shape[0]
        rgb = np.empty((n, 3), dtype=np.uint8)
        hit = np.empty((n, 3), dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self._L.ptr_trace_batch(self._h, dirs.ctypes.data_as(fp), n,
                                            rgb.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            hit.ctypes.data_as(fp) if want_hit else None,
                                            nrm.ctypes.data_as(fp) if want_hit else None))
        return (rgb, hit, nrm) if want_hit else rgb

    # ---- multi-GPU
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId (rank 0 makes it, every rank gets it by the job's own channel)"""
        L = lib()
        buf = (C.c_ubyte * 128)()
        rc = L.ptr_comm_unique_id(buf)
        if rc != OK:
            raise PtrError(rc, L.ptr_last_error(None).decode())
        return bytes(buf)

    def comm_init_rank(self, uid, rank, n_ranks):
        buf = (C.c_ubyte * 128).from_buffer_copy(uid) if uid is not None else None
        self._check(self._L.ptr_comm_init_rank(self._h, buf, rank, n_ranks))

    def gather_strips_device(self, width, height, strip_height, n_frames, d_local, d_full, root=0, frame_stride_px=0, stream=None):
        self._check(self._L.ptr_gather_strips_device(self._h, width, height, strip_height, n_frames, C.c_void_p(d_local),
                                                     frame_stride_px, C.c_void_p(d_full) if d_full else None, root,
                                                     C.c_void_p(stream) if stream else None))

    def place_strips_device(self, part, width, height, n_frames, d_part, d_full, frame_stride_px=0, stream=None):
        self._check(self._L.ptr_place_strips_device(self._h, C.byref(part.c), width, height, n_frames, C.c_void_p(d_part),
                                                    frame_stride_px, C.c_void_p(d_full), C.c_void_p(stream) if stream else None))

    def kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self._L.ptr_get_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats(self):
        s = Stats()
        self._check(self._L.ptr_get_stats(self._h, C.byref(s)))
        return s.as_dict()
