"""Edge cases through the C ABI on the GPU: empty inputs, ragged sizes, frame sequences and cyclic strips on the
kernels that the other files exercise only in their usual shape."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_py as orc          # noqa: E402  (checker only)
import ptr_amd                   # noqa: E402
from ptr_amd import synthetic as syn   # noqa: E402


@pytest.fixture(scope="module")
def rdr():
    r = ptr_amd.Renderer(0)
    yield r
    r.close()


def test_sequence_of_16_sphere_scenes_two_level_kernel(rdr):
    """frames of a sequence on the 13..128-sphere kernel (scalar-loaded records), 1-pixel-wide ragged tiles"""
    torch = pytest.importorskip("torch")
    base = syn.many_spheres(16)
    frames = []
    for f in range(3):
        s = base.copy(); s[0, 1:14] += 0.7 * f; s[1, 1:14] -= 0.3 * f
        frames.append(s)
    rdr.set_camera(None)
    rdr.set_scene(np.stack(frames)); rdr.set_background(syn.BACKGROUND)
    w, h = 97, 45
    out = torch.empty((3, h, w), dtype=torch.int32, device="cuda")
    rdr.render_device(rdr.params(w, h, flags=0), 0, 3, out.data_ptr())
    rdr.sync()
    got = out.cpu().numpy().view(np.uint32)
    for f in range(3):
        want = orc.render(frames[f], syn.BACKGROUND, w, h, flags=0).reshape(h, w)
        assert np.array_equal(got[f], want), f


def test_extension_mode_on_cyclic_strips(rdr, scenes):
    """strips address pixels (and Philox counters) by global row: the parts re-assemble to the whole frame"""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    rdr.set_camera(None); rdr.set_elements(sc["elements"]); rdr.set_background(sc["background"])
    w, h = 128, 70
    full = rdr.render(w, h, flags=ptr_amd.MODE_EXACT_F64, max_depth=3, spp=2, seed=9).reshape(h, w)
    got = np.zeros((h, w), dtype=np.uint32)
    for part in range(3):
        sp = ptr_amd.StripPart(3, part, 8)
        rows = sp.global_rows(h)
        buf = torch.empty((len(rows), w), dtype=torch.int32, device="cuda")
        p = rdr.params(w, h, 0, len(rows), flags=ptr_amd.MODE_EXACT_F64, max_depth=3, spp=2, seed=9)
        rdr.render_strips_device(p, sp, 0, 1, buf.data_ptr())
        rdr.sync()
        got[rows] = buf.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, full)


def test_empty_batches(rdr, scenes):
    sc = scenes["example1"]
    rdr.set_elements(sc["elements"]); rdr.set_background(sc["background"])
    assert rdr.trace_batch_f64(np.zeros((0, 3))).shape == (0, 3)
    rgb, rgb8, hit, nrm = rdr.trace_samples([0, 0, 0], np.zeros((0, 3)))
    assert rgb.shape == (0, 3) and rgb8.shape == (0, 3)


def test_render_cache_passes_on_an_empty_cache(rdr):
    w, h = 17, 9
    cells = (w + 2) * (h + 2)
    view = np.eye(4).T.ravel(); proj = np.eye(4).T.ravel()
    age, item_pixel, el, dep = rdr.rc_reproject([0, 0, 0], view, proj, w, h, np.zeros((0, 3)), np.zeros(0))
    inner = np.zeros((h + 2, w + 2), dtype=bool); inner[1:h + 1, 1:w + 1] = True
    assert (el.reshape(h + 2, w + 2)[inner] == -1).all() and (dep.reshape(h + 2, w + 2)[inner] == 1e6).all()
    assert (el.reshape(h + 2, w + 2)[~inner] == -2).all()                      # border untouched
    out = rdr.rc_shade(w, h, np.full(cells, -1, dtype=np.int32), np.full(cells, 1e6), np.zeros(0), np.zeros((0, 3)))
    assert out["hist"][1, 255] == w * h and out["hist"].sum() == w * h         # every pixel URGENT
    assert out["stats"][0] == 0 and out["stats"][1] == 255 * w * h
    assert (out["color"] == 0xFF000000).all() and (out["priority"] == 0xFFFFFFFF).all()


def test_all_items_without_hits_or_outside(rdr):
    w, h = 16, 16
    view = np.eye(4).T.ravel(); proj = np.eye(4).T.ravel()
    hits = np.array([[np.nan, 0, 0], [5.0, 0, 0], [0.0, 0.0, 0.0], [0.25, 0.25, 0.5]])
    age0 = np.array([1.0, 2.0, 3.0, 4.0])
    age, item_pixel, el, dep = rdr.rc_reproject([0, 0, 0], view, proj, w, h, hits, age0)
    want = orc.rc_reproject([0, 0, 0], view, proj, w, h, hits, age0)
    assert np.array_equal(age, want[0]) and np.array_equal(item_pixel, want[1])
    assert list(age - age0) == [0.0, 2.0, 4.0, 0.0] and item_pixel[3] >= 0


def _small_scene(rng, n, floor):
    soa = np.zeros((12, n))
    soa[0] = rng.uniform(-12, 12, n); soa[1] = rng.uniform(-5, 8, n); soa[2] = rng.uniform(-45, -6, n)
    soa[3] = rng.uniform(0.3, 3.5, n)
    soa[4:7] = rng.uniform(0.0, 1.0, (3, n))
    lights = rng.random(n) < 0.25
    lights[rng.integers(0, n)] = True
    soa[9:12, lights] = rng.uniform(0.2, 2.0, (3, int(lights.sum())))
    if floor:
        soa[0:4, 0] = [0, -1000 - rng.uniform(2, 6), -20, 1000]
        soa[9:12, 0] = 0
    return soa


def test_few_sphere_kernel_random_scenes_ragged_sizes(rdr):
    """1..12 spheres (the SGPR-record kernel: tile culling, own-sphere skip, cone pre-filter of the occluders) at sizes
    that leave ragged tiles; also a scene the own-sphere proof does not cover (a zero-radius element switches it off)."""
    rng = np.random.default_rng(7012)
    rdr.set_camera(None)
    for trial in range(10):
        n = int(rng.integers(1, 13))
        soa = _small_scene(rng, n, trial % 2 == 0)
        if trial == 9:
            soa[3, -1] = 0.0
        bg = rng.uniform(0, 1.5, 3)
        w, h = int(rng.integers(33, 300)), int(rng.integers(17, 200))
        rdr.set_scene(soa); rdr.set_background(bg)
        got = rdr.render(w, h, flags=0)
        want = orc.render(soa, bg, w, h, flags=0, threads=8)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"trial {trial} (n={n}, {w}x{h}): {bad.size} pixels differ, first {bad[:5]}"


def test_few_sphere_kernel_every_element_a_light(rdr):
    """12 elements, all emissive: 12 lights is the most the few-sphere kernel sees, and the shading-only float64 re-do is handed
    one "certainly blocked" bit per light (bits 8, 10, ... 30 of a lane's index register).  Spheres in a tight cluster, so
    that most shadow rays do meet an occluder."""
    rng = np.random.default_rng(7020)
    rdr.set_camera(None)
    for trial in range(3):
        n = 12
        soa = np.zeros((12, n))
        soa[0] = rng.uniform(-5, 5, n); soa[1] = rng.uniform(-3, 4, n); soa[2] = rng.uniform(-26, -14, n)
        soa[3] = rng.uniform(0.8, 2.5, n)
        soa[4:7] = rng.uniform(0.1, 1.0, (3, n))
        soa[9:12] = rng.uniform(0.02, 0.25, (3, n))
        if trial == 2:
            soa[0:4, 0] = [0, -1004, -20, 1000]
        bg = rng.uniform(0, 1.0, 3)
        w, h = 333, 171
        rdr.set_scene(soa); rdr.set_background(bg)
        got = rdr.render(w, h, flags=0)
        st = rdr.stats()
        want = orc.render(soa, bg, w, h, flags=0, threads=8)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"trial {trial}: {bad.size} pixels differ, first {bad[:5]}"
        assert st["reshaded"] > 0, st


def test_few_sphere_kernel_moved_camera_and_inside_a_sphere(rdr):
    """general camera (the culling cone follows it) from several places, one of them inside a big sphere"""
    rng = np.random.default_rng(7013)
    soa = _small_scene(rng, 9, True)
    soa[0:4, 1] = [1.0, 1.0, -3.0, 9.0]          # camera positions below are inside this one
    bg = np.array([0.3, 0.5, 0.9])
    rdr.set_scene(soa); rdr.set_background(bg)
    w, h = 120, 72
    angle = orc.lib().orc_reference_angle()
    for org, look in (((0.5, 1.5, -2.0), (-0.3, -0.2, -1.0)), ((14.0, 6.0, 4.0), (-0.5, -0.25, -1.0)),
                      ((-3.0, 0.5, -60.0), (0.1, 0.0, 1.0))):
        org = np.array(org); fwd = np.array(look); fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        rdr.set_camera(org, right, up, fwd)
        got = rdr.render(w, h, flags=0)
        want = np.empty(w * h, dtype=np.uint32)
        for y in range(h):
            for x in range(w):
                xx = (2 * ((x + 0.5) * (1 / w)) - 1) * angle * (w / h)
                yy = (1 - 2 * ((y + 0.5) * (1 / h))) * angle
                d = xx * right + yy * up + fwd
                l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
                d = d * (1 / np.sqrt(l2))
                want[y * w + x] = orc.pack(orc.trace(soa, bg, org, d))
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"camera at {org}: {bad.size} pixels differ, first {bad[:5]}"
    rdr.set_camera(None)


@pytest.mark.parametrize("n_parts,strip", [(2, 16), (3, 32), (2, 24)])
def test_strips_with_whole_tiles_use_the_culling_path(rdr, scenes, n_parts, strip):
    """strip heights that are multiples of the wave tile's 16 rows keep tile culling on; others switch it off
    (test_gpu_parity.py covers those): the bytes must not depend on it"""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    rdr.set_camera(None)
    rdr.set_elements(sc["elements"]); rdr.set_background(sc["background"])
    w, h = 200, 150
    want = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], w, h, flags=0, threads=8).reshape(h, w)
    full = np.zeros((h, w), dtype=np.uint32)
    max_rows = ptr_amd.StripPart(n_parts, 0, strip).local_rows(h)
    for part in range(n_parts):
        sp = ptr_amd.StripPart(n_parts, part, strip)
        rows = sp.global_rows(h)
        buf = torch.zeros((max_rows, w), dtype=torch.int32, device="cuda")
        p = rdr.params(w, h, 0, len(rows), flags=0)
        rdr.render_strips_device(p, sp, 0, 1, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                                 frame_stride_px=max_rows * w)
        torch.cuda.synchronize()
        full[rows] = buf.cpu().numpy().view(np.uint32)[: len(rows)]
    assert np.array_equal(full, want)


def test_scene_change_between_enqueued_calls_without_a_host_wait(scenes):
    """set_scene, render_device(stream), set_scene, render_device(stream) with no synchronisation in between: the second
    upload is ordered behind the first call's kernels, so neither frame is torn (the upload used to go to the
    context's own stream while the kernels ran on the caller's)"""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    a = ptr_amd.scene_to_soa(sc["elements"])
    b = a.copy(); b[1, 1:5] += 3.0; b[0, 2] = -1.0
    w, h, nf = 1920, 1080, 6                       # long enough that the first call is still running at the second upload
    r = ptr_amd.Renderer(0)
    try:
        r.set_background(sc["background"])
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        out = [torch.empty((nf, h, w), dtype=torch.int32, device="cuda") for _ in range(3)]
        p = r.params(w, h, flags=0)
        r.set_scene(np.stack([a] * nf))
        r.render_device(p, 0, nf, out[0].data_ptr(), stream=s1.cuda_stream)
        r.set_scene(np.stack([b] * nf))
        r.render_device(p, 0, nf, out[1].data_ptr(), stream=s1.cuda_stream)
        r.set_scene(np.stack([a] * nf))            # ... and across streams
        r.render_device(p, 0, nf, out[2].data_ptr(), stream=s2.cuda_stream)
        torch.cuda.synchronize()
        want_a = orc.render(a, sc["background"], w, h, flags=0).reshape(h, w)
        want_b = orc.render(b, sc["background"], w, h, flags=0).reshape(h, w)
        for k, want in enumerate((want_a, want_b, want_a)):
            got = out[k].cpu().numpy().view(np.uint32)
            for f in range(nf):
                assert np.array_equal(got[f], want), (k, f)
        # stats of a call on a caller's stream: ptr_get_stats waits for that call's kernels itself
        r.render_device(r.params(w, h, flags=ptr_amd.FLAG_STATS), 0, 1, out[0].data_ptr(), stream=s2.cuda_stream)
        assert r.stats()["primary"] == w * h
        torch.cuda.synchronize()
    finally:
        r.close()


def test_wide_and_skewed_cameras_render_the_reference_bytes(scenes):
    """fov 120-150 degrees, aspect 16 and a scaled, non-orthogonal basis: the float32 ray of such cameras is outside the
    hybrid mode's error budget, the default mode must still give the float64 bytes"""
    import math
    sc = scenes["example1"]
    soa = ptr_amd.scene_to_soa(sc["elements"])
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(soa); r.set_background(sc["background"])
        cases = [
            dict(w=320, h=180, fov=90, right=(1, 0, 0), up=(0, 1, 0), forward=(0, 0, -1)),        # stays hybrid
            dict(w=320, h=180, fov=150, right=(1, 0, 0), up=(0, 1, 0), forward=(0, 0, -1)),
            dict(w=640, h=40, fov=30, right=(1, 0, 0), up=(0, 1, 0), forward=(0, 0, -1)),         # aspect 16
            dict(w=200, h=120, fov=120, right=(37.0, 0, 3.0), up=(0.5, 41.0, 0), forward=(0.2, -0.1, -39.0)),
        ]
        for c in cases:
            t = math.tan(math.pi * 0.5 * c["fov"] / 180)
            r.set_camera(origin=(0.5, 1.0, 2.0), right=c["right"], up=c["up"], forward=c["forward"], tan_half_fov=t)
            exact = r.render(c["w"], c["h"], flags=ptr_amd.MODE_EXACT_F64)
            hybrid = r.render(c["w"], c["h"], flags=0)
            assert np.array_equal(hybrid, exact), c
            # the oracle's trace on the rays of RayTracer.js:171-174 (same operations, same order, vectorised)
            w, h = c["w"], c["h"]
            R, U, F = (np.array(c[k], dtype=np.float64) for k in ("right", "up", "forward"))
            xs = (2 * ((np.arange(w) + 0.5) * (1 / w)) - 1) * t * (w / h)
            ys = (1 - 2 * ((np.arange(h) + 0.5) * (1 / h))) * t
            xx, yy = np.meshgrid(xs, ys)
            d = [xx * R[k] + yy * U[k] + F[k] for k in range(3)]
            inv = 1 / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
            dirs = np.stack([d[k] * inv for k in range(3)], axis=-1).reshape(-1, 3)
            rgb = orc.trace_batch(soa, np.array(sc["background"], dtype=np.float64), np.tile([0.5, 1.0, 2.0], (w * h, 1)), dirs)[0]
            want = np.array([orc.pack(v) for v in rgb], dtype=np.uint32)
            assert np.array_equal(exact, want), c
    finally:
        r.close()


def test_long_odd_sequence_three_calls_in_a_row(scenes):
    """37 different frames per call, three calls back to back (the queue's counter sets alternate from call to call), the
    last one with the float32 framebuffer as well: every frame against the oracle"""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    base = ptr_amd.scene_to_soa(sc["elements"])
    nf, w, h = 37, 200, 96
    frames = []
    for f in range(nf):
        s = base.copy(); s[0, 1:5] += 0.11 * f; s[1, 2] = -1.0 + 0.07 * f
        frames.append(s)
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(np.stack(frames)); r.set_background(sc["background"])
        st = torch.cuda.Stream()
        out = torch.empty((nf, h, w), dtype=torch.int32, device="cuda")
        f32 = torch.empty((nf, h, w, 4), dtype=torch.float32, device="cuda")
        want = [orc.render(frames[f], sc["background"], w, h, flags=0).reshape(h, w) for f in range(nf)]
        for rep in range(3):
            out.fill_(0)
            r.render_device(r.params(w, h, flags=0), 0, nf, out.data_ptr(), f32.data_ptr() if rep == 2 else None, stream=st.cuda_stream)
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint32)
            for f in range(nf):
                assert np.array_equal(got[f], want[f]), (rep, f)
        assert r.stats()["retraced"] + r.stats()["reshaded"] > 0
        assert float(f32[nf - 1, 0, 0, 3]) == 1.0
    finally:
        r.close()


def test_library_gather_on_one_gpu(scenes):
    """the multi-GPU entry points as far as ONE GPU can take them: a one-rank communicator (nothing is sent), the gather's
    copy step placing the parts of a 3-way partition rendered one after the other, ptr_render_multi on one context, and
    the argument checks.  (The ncclSend / ncclRecv leg needs several GPUs: bench.py --gpus N checks it against
    torch.distributed's gather at start-up.)"""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    soa = ptr_amd.scene_to_soa(sc["elements"])
    moved = soa.copy(); moved[1, 1] = 1.5
    w, h, nf = 200, 77, 2
    want = [orc.render(s, sc["background"], w, h, flags=0).reshape(h, w) for s in (soa, moved)]
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(np.stack([soa, moved])); r.set_background(sc["background"])
        full = torch.zeros((nf, h, w), dtype=torch.int32, device="cuda")
        n_parts = 3
        max_rows = ptr_amd.StripPart(n_parts, 0, 8).local_rows(h)
        for part in range(n_parts):
            sp = ptr_amd.StripPart(n_parts, part, 8)
            rows = sp.local_rows(h)
            local = torch.full((nf, max_rows, w), 0x7F7F7F7F, dtype=torch.int32, device="cuda")
            r.render_strips_device(r.params(w, h, 0, rows, flags=0), sp, 0, nf, local.data_ptr(), frame_stride_px=max_rows * w)
            r.place_strips_device(sp, w, h, nf, local.data_ptr(), full.data_ptr(), frame_stride_px=max_rows * w)
            r.sync()
        got = full.cpu().numpy().view(np.uint32)
        for f in range(nf):
            assert np.array_equal(got[f], want[f]), f
        # a communicator of one rank: the gather is the copy step alone; odd width -> the one-pixel-per-lane form
        with pytest.raises(ptr_amd.PtrError):
            r.gather_strips_device(w, h, 8, nf, full.data_ptr(), full.data_ptr())          # no communicator yet
        r.comm_init_rank(None, 0, 1)
        with pytest.raises(ptr_amd.PtrError):
            r.comm_init_rank(None, 0, 1)                                                   # already a member
        w2 = 199
        local = torch.empty((nf, h, w2), dtype=torch.int32, device="cuda")
        full2 = torch.zeros((nf, h, w2), dtype=torch.int32, device="cuda")
        r.render_device(r.params(w2, h, flags=0), 0, nf, local.data_ptr())
        r.gather_strips_device(w2, h, 8, nf, local.data_ptr(), full2.data_ptr(), root=0)
        r.sync()
        assert torch.equal(local, full2)
        with pytest.raises(ptr_amd.PtrError):
            r.gather_strips_device(w2, h, 8, nf, local.data_ptr(), full2.data_ptr(), root=1)
        one = ptr_amd.render_multi([r], w, h)                                               # frame 0, whole, host array
        assert np.array_equal(one.reshape(h, w), want[0])
    finally:
        r.close()
    a, b = ptr_amd.Renderer(0), ptr_amd.Renderer(0)
    try:
        with pytest.raises(ptr_amd.PtrError) as e:                                         # two ranks on one GPU
            ptr_amd.comm_init_all([a, b])
        assert "same GPU" in str(e.value)
        ptr_amd.comm_init_all([a])
        a.set_scene(soa); a.set_background(sc["background"])
        assert np.array_equal(ptr_amd.render_multi([a], w, h).reshape(h, w), want[0])
    finally:
        a.close(); b.close()


def test_host_buffer_copies_pageable_pinned_and_sizes(scenes):
    """ptr_render's way back to the host: a page-locked destination takes one DMA, a pageable one is filled through the
    context's two page-locked halves in pieces of at most 1 MiB (copy_to_host).  Same bytes as the device-buffer entry for
    frames below the staging threshold, of exactly one piece, of several pieces with a ragged last one, for the float32 frame
    (16 bytes per pixel: many pieces) and for a destination that does not start on a 16-byte boundary."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    r = ptr_amd.Renderer(0)
    try:
        r.set_elements(sc["elements"]); r.set_background(sc["background"])
        for (w, h) in ((64, 48), (256, 256), (512, 512), (1920, 1080), (1001, 517)):
            p = r.params(w, h, flags=0)
            dev8 = torch.zeros((h, w), dtype=torch.int32, device="cuda")
            devf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
            r.render_device(p, 0, 1, dev8.data_ptr(), devf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want8 = dev8.cpu().numpy().view(np.uint32).ravel(); wantf = devf.cpu().numpy()
            got8, gotf = r.render(w, h, flags=0, want_f32=True)                       # pageable numpy buffers
            assert np.array_equal(got8, want8) and np.array_equal(gotf, wantf), (w, h)
            pin8 = torch.zeros(w * h, dtype=torch.int32).pin_memory()                 # page-locked: the direct copy
            r._check(r._L.ptr_render(r._h, C.byref(p), C.cast(pin8.data_ptr(), C.POINTER(C.c_uint32)), None))
            assert np.array_equal(pin8.numpy().view(np.uint32), want8), (w, h)
            raw = np.zeros(w * h * 4 + 64, dtype=np.uint8)                            # pageable, 4 bytes off a 16-byte boundary
            off = (-raw.ctypes.data) % 16 + 4
            view = raw[off:off + w * h * 4]
            r._check(r._L.ptr_render(r._h, C.byref(p), C.cast(view.ctypes.data, C.POINTER(C.c_uint32)), None))
            assert np.array_equal(view.view(np.uint32), want8), (w, h)
            assert not raw[:off].any() and not raw[off + w * h * 4:].any()
    finally:
        r.close()
