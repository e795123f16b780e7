"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against
(a) the committed golden vectors made by the unmodified reference under Node and
(b) the CPU oracle on the same seeded inputs.

Bars: float64 ("exact") and hybrid modes must reproduce the reference's bytes; the float32-only
mode must stay within max per-channel error <= 1/255 (the tolerance BASELINE.json's north_star
states) except for a bounded number of pixels on discontinuities, which is reported."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_py as orc          # noqa: E402  (checker only)
import ptr_amd                   # noqa: E402

Q = ptr_amd.QUIRK_ROW_SHIFT
MODES = {"exact": ptr_amd.MODE_EXACT_F64, "hybrid": 0}


@pytest.fixture(scope="module")
def rdr():
    r = ptr_amd.Renderer(0)
    yield r
    r.close()


def _load(r, sc):
    r.set_camera(None)
    r.set_elements(sc["elements"])
    r.set_background(sc["background"])


def _frame(golden_dir, rel):
    return np.fromfile(os.path.join(golden_dir, rel), dtype=np.uint32)


def _channel_diff(a, b):
    a8 = a.view(np.uint8).reshape(-1, 4).astype(np.int16)
    b8 = b.view(np.uint8).reshape(-1, 4).astype(np.int16)
    return np.abs(a8 - b8).max(axis=1)


@pytest.mark.parametrize("mode", ["exact", "hybrid"])
def test_small_golden_frames_bit_exact(rdr, golden_dir, golden_index, scenes, mode):
    for key, meta in golden_index["frames"].items():
        sc = scenes[key.split("@")[0]]
        _load(rdr, sc)
        got = rdr.render(meta["width"], meta["height"], flags=Q | MODES[mode])
        want = _frame(golden_dir, meta["file"])
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"{mode} {key}: {bad.size} pixels differ, first {bad[:5]}"


@pytest.mark.parametrize("mode", ["exact", "hybrid"])
@pytest.mark.parametrize("key", ["example1@640x480", "example1@1920x1080", "example1@3840x2160",
                                 "anim_t1234@1920x1080", "random_5@640x480"])
def test_big_frame_hashes(rdr, golden_index, scenes, key, mode):
    meta = golden_index["hashes"][key]
    sc = scenes[key.split("@")[0]]
    _load(rdr, sc)
    w, h = meta["width"], meta["height"]
    got = rdr.render(w, h, flags=Q | MODES[mode])
    for x, y, v in meta["probes"]:
        assert int(got[y * w + x]) == v, (x, y)
    assert hashlib.sha256(got.tobytes()).hexdigest() == meta["sha256"]
    assert not got[(h - 1) * w:].any()          # F5: last stored row untouched
    if mode == "hybrid":
        st = rdr.stats()
        assert 0 < st["retraced"] + st["reshaded"] < 0.15 * w * h, st
        # doubts about the 8-bit rounding alone take the shading-only float64 re-do (most of them on the example scene)
        assert st["reshaded"] > 0 and st["retraced"] > 0, st
        if key.startswith("example1"):
            assert st["reshaded"] > st["retraced"], st


MORE_HASHES = ["random_0@640x480", "random_1@640x480", "random_2@640x480", "random_3@640x480", "random_4@640x480", "random_6@640x480",
               "random_7@640x480", "inside_sphere@640x480", "odd_colours@640x480", "coincident@640x480", "light_below_horizon@640x480",
               "occluder_behind_light@640x480", "no_lights@320x240", "example1_permuted@800x600", "anim_t98765@1280x720",
               "example1@1001x563", "random_7@1279x719"]


@pytest.mark.parametrize("mode", ["exact", "hybrid"])
@pytest.mark.parametrize("key", MORE_HASHES)
def test_more_reference_hashes(rdr, golden_index, scenes, key, mode):
    """hashes alone of the other golden scenes at larger and odd sizes (the unmodified reference under Node, oracle/gen_golden.mjs)"""
    meta = golden_index["hashes"][key]
    _load(rdr, scenes[key.split("@")[0]])
    w, h = meta["width"], meta["height"]
    got = rdr.render(w, h, flags=Q | MODES[mode])
    for x, y, v in meta["probes"]:
        assert int(got[y * w + x]) == v, (x, y)
    assert hashlib.sha256(got.tobytes()).hexdigest() == meta["sha256"]


@pytest.mark.parametrize("mode", ["exact", "hybrid"])
def test_banded_render(rdr, golden_dir, golden_index, scenes, mode):
    for key, meta in golden_index["bands"].items():
        sc = scenes[key.split("@")[0]]
        _load(rdr, sc)
        w, h, jobs, band = meta["width"], meta["height"], meta["jobs"], meta["band"]
        parts = [rdr.render(w, h, i * band, band, flags=Q | MODES[mode]) for i in range(jobs)]
        assert np.array_equal(np.concatenate(parts), _frame(golden_dir, meta["file"])), key


def test_unshifted_layout_matches_oracle(rdr, scenes):
    sc = scenes["example1"]
    _load(rdr, sc)
    soa = orc.scene_to_soa(sc["elements"])
    for mode in MODES.values():
        got = rdr.render(200, 150, 10, 100, flags=mode)
        want = orc.render(soa, sc["background"], 200, 150, 10, 100, flags=0)
        assert np.array_equal(got, want)


def test_f32_only_within_tolerance(rdr, golden_index, scenes):
    """north_star tolerance: max per-channel error <= 1/255.  float32 alone cannot promise it on
    discontinuities (silhouettes, shadow edges): count and bound those pixels."""
    total = worst = 0
    for name, (w, h) in {"example1": (640, 480), "anim_t1234": (640, 480), "random_5": (640, 480),
                         "random_7": (320, 240)}.items():
        sc = scenes[name]
        _load(rdr, sc)
        got = rdr.render(w, h, flags=ptr_amd.MODE_F32_ONLY)
        want = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], w, h, flags=0, threads=8)
        d = _channel_diff(got, want)
        over = int((d > 1).sum())
        total += w * h
        worst += over
        assert over <= max(8, 2e-4 * w * h), f"{name}: {over} pixels off by more than 1/255"
    print(f"f32-only: {worst} of {total} pixels beyond 1/255")


def test_trace_batch_f64_bit_exact(rdr, golden_dir, golden_index, scenes):
    for name, meta in golden_index["trace"].items():
        rec = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.float64).reshape(-1, 9)
        _load(rdr, scenes[name])
        got = rdr.trace_batch_f64(rec[:, 3:6], origins=rec[:, 0:3])
        assert got.tobytes() == np.ascontiguousarray(rec[:, 6:9]).tobytes(), name


def test_trace_batch_hit_and_normal_match_oracle(rdr, golden_dir, golden_index, scenes):
    rec = np.fromfile(os.path.join(golden_dir, golden_index["trace"]["example1"]["file"]),
                      dtype=np.float64).reshape(-1, 9)
    sc = scenes["example1"]
    _load(rdr, sc)
    rgb, el, hit, nrm = rdr.trace_batch_f64(rec[:, 3:6], origins=rec[:, 0:3], want_hit=True)
    orgb, oel, opt, onr, _ = orc.trace_batch(orc.scene_to_soa(sc["elements"]), sc["background"],
                                             rec[:, 0:3], rec[:, 3:6])
    assert np.array_equal(el, oel)
    assert hit.tobytes() == opt.tobytes() and nrm.tobytes() == onr.tobytes()
    assert rgb.tobytes() == orgb.tobytes()
    assert (el >= 0).sum() > 100 and (el < 0).sum() > 100


def test_request_pixel_bytes(rdr, golden_dir, golden_index, scenes):
    meta = golden_index["request_pixel"]["example1"]
    rec = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.float64).reshape(-1, 6)
    _load(rdr, scenes["example1"])
    # the float entry point rounds directions to float32 first: feed float32-exact directions
    d32 = rec[:, 0:3].astype(np.float32)
    got = rdr.trace_batch(d32)
    sc = scenes["example1"]
    soa = orc.scene_to_soa(sc["elements"])
    want = np.array([orc.request_pixel(soa, sc["background"], d.astype(np.float64)) for d in d32])
    assert np.array_equal(got.astype(np.float64), want)


def test_stats_match_reference_ray_mix(rdr, scenes):
    # instrumented reference at 640x480
    _load(rdr, scenes["example1"])
    rdr.render(640, 480, flags=Q | ptr_amd.FLAG_STATS)
    st = rdr.stats()
    assert st["primary"] == 307200 and st["hits"] == 240687 and st["shadow_rays"] == 481374
    assert st["exit_tca"] + st["exit_d2"] + st["exit_hit"] == 4371301


def test_random_scenes_hybrid_vs_oracle(rdr):
    rng = np.random.default_rng(20261111)
    retraced = pixels = 0
    for trial in range(12):
        n = int(rng.integers(1, 40))
        soa = np.zeros((12, n))
        soa[0] = rng.uniform(-15, 15, n); soa[1] = rng.uniform(-6, 10, n); soa[2] = rng.uniform(-50, -6, n)
        soa[3] = rng.uniform(0.2, 3.0, n)
        soa[4:7] = rng.uniform(0.0, 1.0, (3, n))
        lights = rng.random(n) < 0.15
        lights[rng.integers(0, n)] = True
        soa[9:12, lights] = rng.uniform(0.2, 2.0, (3, int(lights.sum())))
        if trial % 3 == 0:                     # a floor
            soa[0:4, 0] = [0, -1000 - rng.uniform(2, 6), -20, 1000]
            soa[9:12, 0] = 0
        bg = rng.uniform(0, 1.5, 3)
        w, h = 256, 192
        rdr.set_camera(None)
        rdr.set_scene(soa)
        rdr.set_background(bg)
        got = rdr.render(w, h, flags=0)
        want = orc.render(soa, bg, w, h, flags=0, threads=8)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"trial {trial} (n={n}): {bad.size} pixels differ, first {bad[:5]}"
        retraced += rdr.stats()["retraced"]
        pixels += w * h
    print(f"hybrid: {retraced} of {pixels} pixels re-traced in float64 ({100.0 * retraced / pixels:.2f}%)")
    assert retraced < 0.2 * pixels


def test_general_camera_vs_oracle_rays(rdr, scenes):
    """A moved / rotated camera (extension of the reference's fixed one, F4): the exact kernels
    must agree with oracle traces of the same rays."""
    sc = scenes["example1"]
    _load(rdr, sc)
    org = np.array([3.0, 2.0, 5.0])
    fwd = np.array([-0.2, -0.1, -1.0]); fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    rdr.set_camera(org, right, up, fwd)
    w, h = 96, 64
    got = rdr.render(w, h, flags=ptr_amd.MODE_EXACT_F64)
    got_h = rdr.render(w, h, flags=0)
    got_m = rdr.render(w, h, flags=ptr_amd.DEBUG_FORCE_MANY)          # the two-level kernel's packed ray set-up
    angle = orc.lib().orc_reference_angle()
    soa = orc.scene_to_soa(sc["elements"])
    want = np.empty(w * h, dtype=np.uint32)
    for y in range(h):
        for x in range(w):
            xx = (2 * ((x + 0.5) * (1 / w)) - 1) * angle * (w / h)
            yy = (1 - 2 * ((y + 0.5) * (1 / h))) * angle
            d = xx * right + yy * up + fwd
            l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
            d = d * (1 / np.sqrt(l2))
            want[y * w + x] = orc.pack(orc.trace(soa, sc["background"], org, d))
    rdr.set_camera(None)
    assert np.array_equal(got, want)
    assert np.array_equal(got_h, want)
    assert np.array_equal(got_m, want)


def test_float32_framebuffer(rdr, scenes):
    sc = scenes["example1"]
    _load(rdr, sc)
    w, h = 160, 120
    want = orc.render_rgb64(orc.scene_to_soa(sc["elements"]), sc["background"], w, h)
    for flags in (ptr_amd.MODE_EXACT_F64, 0, ptr_amd.MODE_F32_ONLY):
        _, f32 = rdr.render(w, h, flags=flags, want_f32=True)
        assert np.all(f32[..., 3] == 1.0)
        err = np.abs(f32[..., :3].astype(np.float64) - want)
        if flags == ptr_amd.MODE_EXACT_F64:
            assert np.array_equal(f32[..., :3], want.astype(np.float32))
        else:
            # tolerance: 1e-4 absolute on all but discontinuity pixels (<= 0.1 % of the frame)
            assert (err.max(axis=2) > 1e-4).mean() <= 1e-3


def test_sequence_render_device(rdr, scenes):
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    base = orc.scene_to_soa(sc["elements"])
    frames = []
    for f in range(5):
        s = base.copy()
        s[1, 1] = 5 * np.sin(f * 0.4)          # example2.js:101 style motion
        s[0, 5] = 10 * np.sin(f * 0.2)
        frames.append(s)
    seq = np.stack(frames)
    rdr.set_camera(None)
    rdr.set_scene(seq)
    rdr.set_background(sc["background"])
    w, h = 320, 240
    out = torch.zeros((3, h, w), dtype=torch.int32, device="cuda")
    p = rdr.params(w, h, flags=0)
    rdr.render_device(p, 1, 3, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    for k in range(3):
        want = orc.render(frames[1 + k], sc["background"], w, h, flags=0, threads=8)
        assert np.array_equal(got[k].ravel(), want), k


def test_scene_is_reread_every_call(rdr, scenes):
    # example2.js mutates sphere centres between frames: nothing may be cached across set_scene
    sc = scenes["example1"]
    _load(rdr, sc)
    a = rdr.render(64, 48, flags=Q)
    rdr.set_elements(scenes["anim_t1234"]["elements"])
    b = rdr.render(64, 48, flags=Q)
    _load(rdr, sc)
    c = rdr.render(64, 48, flags=Q)
    assert np.array_equal(a, c) and not np.array_equal(a, b)


def test_errors(scenes):
    r = ptr_amd.Renderer(0)
    try:
        with pytest.raises(ptr_amd.PtrError) as e:
            r.render(8, 8)
        assert e.value.code == ptr_amd.ERR_NO_SCENE
        r.set_elements(scenes["example1"]["elements"])
        with pytest.raises(ptr_amd.PtrError) as e:
            r.render(0, 8)
        assert e.value.code == ptr_amd.ERR_INVALID
        with pytest.raises(ptr_amd.PtrError) as e:
            r.render(8, 8, flags=ptr_amd.MODE_EXACT_F64 | ptr_amd.MODE_F32_ONLY)
        assert e.value.code == ptr_amd.ERR_INVALID
        p = r.params(8, 8, max_depth=17)
        out = np.empty(64, dtype=np.uint32)
        import ctypes as C
        rc = ptr_amd.lib().ptr_render(r._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_uint32)), None)
        assert rc == ptr_amd.ERR_INVALID
        p = r.params(1 << 24, 8, 0, 1)            # the kernels index pixels with 24-bit multiplies: refused before any launch
        rc = ptr_amd.lib().ptr_render(r._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_uint32)), None)
        assert rc == ptr_amd.ERR_INVALID and b"width" in ptr_amd.lib().ptr_last_error(r._h)
        with pytest.raises(ptr_amd.PtrError):
            ptr_amd.Renderer(99)
    finally:
        r.close()


@pytest.mark.parametrize("n_parts,strip", [(2, 8), (4, 8), (8, 8), (3, 5)])
def test_cyclic_strips_reassemble_to_the_frame(rdr, scenes, n_parts, strip):
    """Multi-GPU sharding unit: the strips of all parts, re-interleaved, equal the whole frame
    (the same bytes for any number of parts)."""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    _load(rdr, sc)
    w, h = 320, 180
    want = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], w, h, flags=0, threads=8).reshape(h, w)
    full = np.zeros((h, w), dtype=np.uint32)
    max_rows = ptr_amd.StripPart(n_parts, 0, strip).local_rows(h)
    seen = 0
    for part in range(n_parts):
        sp = ptr_amd.StripPart(n_parts, part, strip)
        rows = sp.global_rows(h)
        assert sp.local_rows(h) == len(rows)
        buf = torch.zeros((max_rows, w), dtype=torch.int32, device="cuda")
        p = rdr.params(w, h, 0, len(rows), flags=0)
        rdr.render_strips_device(p, sp, 0, 1, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                                 frame_stride_px=max_rows * w)
        torch.cuda.synchronize()
        full[rows] = buf.cpu().numpy().view(np.uint32)[: len(rows)]
        seen += len(rows)
    assert seen == h
    assert np.array_equal(full, want)
