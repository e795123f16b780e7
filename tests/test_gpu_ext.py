"""GPU tests of the EXTENSION modes (max_depth > 0: reflection/refraction; spp > 1: Philox jitter),
through the C ABI.  The reference implements neither, so the bar is the float64 specification
(oracle/ptr_oracle.c orc_render_ext, self-pinned by tests/golden/ext — "parity unpinned"):
  default mode and PTR_MODE_EXACT_F64: identical bytes;
  PTR_MODE_F32_ONLY (the float32 throughput form): per-channel error <= 1/255 except on at most 0.5 % of the pixels
  (discontinuities: silhouettes, shadow edges and their mirror images), and PSNR >= 50 dB."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_py as orc          # noqa: E402  (checker only)
import ptr_amd                   # noqa: E402

X = ptr_amd.MODE_EXACT_F64


@pytest.fixture(scope="module")
def rdr():
    r = ptr_amd.Renderer(0)
    yield r
    r.close()


def _load(r, sc):
    r.set_camera(None)
    r.set_elements(sc["elements"])
    r.set_background(sc["background"])


def _index(golden_dir):
    with open(os.path.join(golden_dir, "ext", "index.json")) as f:
        return json.load(f)["cases"]


def test_exact_matches_frozen_vectors(rdr, golden_dir, scenes):
    for key, c in _index(golden_dir).items():
        if c["scene"] not in scenes:
            continue                      # BASELINE configs 2 / 3 on the synthetic scene: tests of their own below
        _load(rdr, scenes[c["scene"]])
        got = rdr.render(c["width"], c["height"], flags=X, max_depth=c["max_depth"], spp=c["spp"], seed=c["seed"])
        assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"], key
        if "file" in c:
            want = np.fromfile(os.path.join(golden_dir, c["file"]), dtype=np.uint32)
            assert np.array_equal(got, want), key


@pytest.mark.parametrize("name,depth,spp", [("random_2", 5, 1), ("random_4", 4, 2), ("random_7", 2, 1),
                                            ("light_below_horizon", 5, 3), ("empty", 5, 2),
                                            ("coincident", 5, 1), ("odd_colours", 2, 2)])
def test_exact_matches_specification(rdr, scenes, name, depth, spp):
    sc = scenes[name]
    _load(rdr, sc)
    got = rdr.render(128, 96, flags=X, max_depth=depth, spp=spp, seed=42)
    want = orc.render_ext(orc.scene_to_soa(sc["elements"]), sc["background"], 128, 96,
                          max_depth=depth, spp=spp, seed=42)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{name}: {bad.size} pixels differ, first {bad[:5]}"


def test_exact_bands_tile_the_frame(rdr, scenes):
    sc = scenes["example1"]
    _load(rdr, sc)
    full = rdr.render(200, 150, flags=X, max_depth=5, spp=2, seed=3).reshape(150, 200)
    parts = [rdr.render(200, 150, y0, h, flags=X, max_depth=5, spp=2, seed=3).reshape(h, 200)
             for y0, h in ((0, 1), (1, 70), (71, 79))]
    assert np.array_equal(full, np.vstack(parts))


def test_exact_general_camera_rgb32f(rdr, scenes):
    # out_rgba32f carries the unclamped mean colour; moved camera == moved scene
    sc = scenes["example1"]
    soa = orc.scene_to_soa(sc["elements"])
    _load(rdr, sc)
    rdr.set_camera(origin=(0.5, 0.25, 2.0))
    got, f32 = rdr.render(96, 64, flags=X, max_depth=3, want_f32=True)
    rdr.set_camera(None)
    _, ref64 = orc.render_ext(soa, sc["background"], 96, 64, max_depth=3, want_rgb64=True)
    assert np.isfinite(f32).all() and f32[..., 3].min() == 1.0
    moved = soa.copy()
    moved[0] -= 0.5; moved[1] -= 0.25; moved[2] -= 2.0
    _, want64 = orc.render_ext(moved, sc["background"], 96, 64, max_depth=3, want_rgb64=True)
    assert np.abs(f32[..., :3] - want64).max() < 1e-3          # translation in f64: tiny rounding only
    assert not np.allclose(want64, ref64)


def _psnr(a, b):
    a8 = a.view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    b8 = b.view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    mse = ((a8 - b8) ** 2).mean()
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


@pytest.mark.parametrize("flags", [0, ptr_amd.MODE_F32_ONLY])
@pytest.mark.parametrize("name,w,h,depth,spp", [("example1", 640, 480, 5, 1), ("example1", 320, 240, 5, 4),
                                                ("random_3", 320, 240, 5, 1), ("random_5", 320, 240, 3, 2)])
def test_float32_within_tolerance(rdr, scenes, name, w, h, depth, spp, flags):
    sc = scenes[name]
    _load(rdr, sc)
    got = rdr.render(w, h, flags=flags, max_depth=depth, spp=spp, seed=11)
    want = orc.render_ext(orc.scene_to_soa(sc["elements"]), sc["background"], w, h,
                          max_depth=depth, spp=spp, seed=11)
    d = np.abs(got.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
    off = int((d > 1).sum())
    print(f"{name} d{depth} s{spp}: {int((d > 0).sum())} px differ, {off} by more than 1/255, "
          f"PSNR {_psnr(got, want):.1f} dB")
    if flags == 0:
        assert int((d > 0).sum()) == 0        # the default mode is the float64 specification: <= 1/255 everywhere (0, in fact)
    else:
        assert off <= 0.005 * d.size          # the float32 throughput form, asked for by name
        assert _psnr(got, want) >= 50.0


def test_full_hd_depth5_hash(rdr, golden_dir, scenes):
    c = _index(golden_dir)["example1@1920x1080_d5_s1_k0"]
    _load(rdr, scenes["example1"])
    got = rdr.render(1920, 1080, flags=X, max_depth=5)
    assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"]


def _rmse(a, b):
    a8 = a.view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    b8 = b.view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
    return float(np.sqrt(((a8 - b8) ** 2).mean()))


def test_baseline_config2_1080p_64spp(rdr, golden_dir):
    """BASELINE.json configs[2]: 1920x1080, 64 Philox samples per pixel, depth 5, diffuse + reflective spheres
    (synthetic.config3_scene, seed 0x5EED1234).  float64 mode: the frozen hash of the CPU specification; float32 modes:
    SURVEY.md 8(d)'s tolerance against that frame, PSNR >= 50 dB and RMSE <= 0.5/255 (measured values printed)."""
    from ptr_amd import synthetic as syn
    c = _index(golden_dir)["config2@1920x1080_d5_s64"]
    rdr.set_camera(None); rdr.set_scene(syn.config3_scene()); rdr.set_background(syn.BACKGROUND)
    kw = dict(max_depth=c["max_depth"], spp=c["spp"], seed=c["seed"])
    exact = rdr.render(c["width"], c["height"], flags=X, **kw)
    assert hashlib.sha256(exact.tobytes()).hexdigest() == c["sha256"]
    assert np.array_equal(rdr.render(c["width"], c["height"], flags=0, **kw), exact)      # the default mode is the specification
    for flags in (ptr_amd.MODE_F32_ONLY,):
        got = rdr.render(c["width"], c["height"], flags=flags, **kw)
        d = np.abs(got.view(np.uint8).astype(np.int16) - exact.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
        print(f"config 2, flags {flags}: PSNR {_psnr(got, exact):.2f} dB, RMSE {_rmse(got, exact):.4f}/255, "
              f"{int((d > 1).sum())} of {d.size} pixels beyond 1/255")
        assert _psnr(got, exact) >= 50.0 and _rmse(got, exact) <= 0.5


def test_baseline_config3_one_part_of_4k_256spp(rdr, golden_dir):
    """BASELINE.json configs[3]: 3840x2160, 256 samples per pixel, depth 5, sharded into cyclic 8-row strips over 8 GPUs:
    the rows of part 3 of 8 (what rank 3 renders), float64 mode, against the frozen hash of the CPU specification's rows;
    float32 against the float64 rows at config 2's tolerance"""
    torch = pytest.importorskip("torch")
    from ptr_amd import synthetic as syn
    c = _index(golden_dir)["config3_part3of8@3840x2160_d5_s256"]
    rdr.set_camera(None); rdr.set_scene(syn.config3_scene()); rdr.set_background(syn.BACKGROUND)
    n_parts, part, sh = c["strips"]
    sp = ptr_amd.StripPart(n_parts, part, sh)
    rows = sp.local_rows(c["height"])
    assert rows * c["width"] == c["pixels"]
    buf = torch.empty((rows, c["width"]), dtype=torch.int32, device="cuda")
    frames = {}
    F = ptr_amd.MODE_F32_ONLY
    for flags in (X, F):
        p = rdr.params(c["width"], c["height"], 0, rows, flags=flags, max_depth=c["max_depth"], spp=c["spp"], seed=c["seed"])
        rdr.render_strips_device(p, sp, 0, 1, buf.data_ptr())
        rdr.sync()
        frames[flags] = buf.cpu().numpy().view(np.uint32).ravel().copy()
    assert hashlib.sha256(frames[X].tobytes()).hexdigest() == c["sha256"]
    print(f"config 3 part: float32 PSNR {_psnr(frames[F], frames[X]):.2f} dB, RMSE {_rmse(frames[F], frames[X]):.4f}/255")
    assert _psnr(frames[F], frames[X]) >= 50.0 and _rmse(frames[F], frames[X]) <= 0.5


def test_planes_and_phong_against_the_specification(rdr, golden_dir):
    """ray-plane primitive + Phong terms (north_star names both; the reference has neither: self-pinned like the bounce
    loop).  float64 (default mode): the frozen frames of the CPU specification; float32 by name: PSNR >= 50 dB; the
    extension data is cleared again and the reference path is back."""
    from ptr_amd import synthetic as syn
    soa, planes, phong = syn.planes_scene()
    idx = _index(golden_dir)
    rdr.set_camera(None); rdr.set_scene(soa); rdr.set_background(syn.BACKGROUND)
    rdr.set_extension(planes, phong)
    try:
        for key in ("planes_phong@160x120_d0_s1", "planes_phong@160x120_d3_s2", "planes_phong@640x480_d5_s1"):
            c = idx[key]
            kw = dict(max_depth=c["max_depth"], spp=c["spp"], seed=c["seed"])
            got = rdr.render(c["width"], c["height"], flags=0, **kw)
            assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"], key
            if "file" in c:
                assert np.array_equal(got, np.fromfile(os.path.join(golden_dir, c["file"]), dtype=np.uint32)), key
            f32 = rdr.render(c["width"], c["height"], flags=ptr_amd.MODE_F32_ONLY, **kw)
            print(f"{key}: float32 PSNR {_psnr(f32, got):.1f} dB")
            assert _psnr(f32, got) >= 50.0
        # the picture really has a floor plane and highlights: it differs from the same spheres without the extension data
        with_ext = rdr.render(160, 120, flags=0)
        # random planes / Phong terms against the CPU specification, straight
        rng = np.random.default_rng(5)
        for t in range(3):
            pl = planes.copy(); pl[1, 0] = -4.0 - t; pl[6:9, 1] = rng.uniform(0.2, 1.0, 3)
            ph = phong.copy(); ph[0] = rng.uniform(0, 0.8, ph.shape[1]); ph[1] = rng.integers(1, 200, ph.shape[1])
            rdr.set_extension(pl, ph)
            got = rdr.render(96, 64, flags=0, max_depth=2 + t, spp=1 + t, seed=t)
            want = orc.render_ext2(soa, syn.BACKGROUND, 96, 64, planes=pl, phong=ph, max_depth=2 + t, spp=1 + t, seed=t)
            assert np.array_equal(got, want), t
        with pytest.raises(ptr_amd.PtrError):
            rdr.trace_batch_f64(np.array([[0.0, 0.0, -1.0]]))            # single rays: reference semantics only
        with pytest.raises(ptr_amd.PtrError):
            rdr.set_extension(planes, phong[:, :4]); rdr.render(32, 32, flags=0)      # Phong terms for the wrong element count
    finally:
        rdr.set_extension(None, None)
    plain = rdr.render(160, 120, flags=0)
    assert np.array_equal(plain, orc.render(soa, syn.BACKGROUND, 160, 120, flags=0))   # the reference path again
    assert not np.array_equal(plain, with_ext)


def test_work_counters_match_the_specification(rdr, scenes):
    for name, depth, spp in (("example1", 5, 1), ("random_3", 3, 2), ("example1", 0, 1)):
        sc = scenes[name]
        _load(rdr, sc)
        if depth == 0 and spp == 1:
            continue                      # the reference path has its own counter test
        rdr.render(160, 120, flags=ptr_amd.FLAG_STATS, max_depth=depth, spp=spp, seed=4)
        got = rdr.stats()
        want = orc.count_ext(orc.scene_to_soa(sc["elements"]), sc["background"], 160, 120, max_depth=depth, spp=spp, seed=4)
        for k in ("primary", "hits", "shadow_rays", "exit_tca", "exit_d2", "exit_hit", "secondary"):
            assert got[k] == want[k], (name, k, got, want)


def test_rejections(rdr, scenes):
    _load(rdr, scenes["example1"])
    for kw, code in ((dict(max_depth=17), ptr_amd.ERR_INVALID), (dict(max_depth=-1), ptr_amd.ERR_INVALID),
                     (dict(spp=0), ptr_amd.ERR_INVALID),
                     (dict(max_depth=2, flags=ptr_amd.QUIRK_ROW_SHIFT), ptr_amd.ERR_INVALID)):
        kw.setdefault("flags", 0)
        with pytest.raises(ptr_amd.PtrError) as e:
            rdr.render(64, 48, **kw)
        assert e.value.code == code, kw


def test_random_scenes_against_the_specification():
    """tools/soak_ext.py, bounded: 120 random scenes (1-39 spheres, reflective / transparent / nested / a huge floor; depth 0-7,
    1-5 samples, odd frame sizes; a third with planes and Phong terms): the float64 kernel (default mode) equals the CPU specification on every pixel.  The float32
    form on the same scenes: at most 0.05 % of the pixels beyond 1/255 (measured over 3 000 scenes, a third of them with planes
    and Phong terms: 0.0058 % since the hit point is put back onto its sphere, 0.011 - 0.015 % before)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak_ext
    px, bad, beyond, worst = soak_ext.soak(120, 11, verbose=False)
    print(f"{px} pixels: float64 differs on {bad}; float32 beyond 1/255 on {beyond}, worst PSNR {worst:.1f} dB")
    assert bad == 0
    assert beyond <= 5e-4 * px


def test_float32_form_with_turned_cameras_bands_and_strips(rdr, scenes):
    """The float32 bounce loop culls the camera ray's sphere tests per 8 x 8 tile (cone around the tile's centre ray): a wrong
    cone would make spheres vanish from whole tiles.  Against the float64 kernel (pinned to the specification above) with cameras
    that are moved, turned and wide, in a band that does not start on a tile row, and in cyclic strips."""
    import math
    sc = scenes["example1"]
    _load(rdr, sc)
    cams = [dict(origin=(0.0, 0.0, 0.0)),
            dict(origin=(3.0, 1.5, 4.0)),
            dict(origin=(-6.0, 2.0, -8.0), right=(math.cos(0.6), 0.0, math.sin(0.6)), up=(0.0, 1.0, 0.0),
                 forward=(math.sin(0.6), 0.0, -math.cos(0.6))),
            dict(origin=(0.0, 12.0, -14.0), right=(1.0, 0.0, 0.0), up=(0.0, math.cos(1.0), -math.sin(1.0)),
                 forward=(0.0, -math.sin(1.0), -math.cos(1.0)), tan_half_fov=math.tan(math.radians(50)))]
    try:
        for cam in cams:
            rdr.set_camera(**cam)
            for (w, h, y0, rows, depth, spp) in ((333, 201, 0, 201, 4, 1), (256, 160, 13, 90, 3, 3)):
                f32 = rdr.render(w, h, y0, rows, flags=ptr_amd.MODE_F32_ONLY, max_depth=depth, spp=spp, seed=5)
                f64 = rdr.render(w, h, y0, rows, flags=X, max_depth=depth, spp=spp, seed=5)
                d = np.abs(f32.view(np.uint8).astype(np.int16) - f64.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
                assert int((d > 1).sum()) <= 0.005 * d.size and _psnr(f32, f64) >= 45.0, (cam, w, h, int((d > 1).sum()), _psnr(f32, f64))
            # cyclic strips of 8 rows (whole tiles: culled) and of 5 rows (not whole tiles: no culling), 3 parts
            import torch
            w, h = 200, 120
            whole = rdr.render(w, h, flags=ptr_amd.MODE_F32_ONLY, max_depth=3).reshape(h, w)
            for strip in (8, 5):
                for part_no in range(3):
                    part = ptr_amd.StripPart(3, part_no, strip)
                    rows = part.local_rows(h)
                    buf = torch.zeros((rows, w), dtype=torch.int32, device="cuda")
                    p = rdr.params(w, h, 0, rows, flags=ptr_amd.MODE_F32_ONLY, max_depth=3)
                    rdr.render_strips_device(p, part, 0, 1, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    got = buf.cpu().numpy().view(np.uint32)
                    assert np.array_equal(got, whole[part.global_rows(h)]), (cam, strip, part_no)
    finally:
        rdr.set_camera(None)
        rdr.set_extension(None, None)


def test_float32_sample_blocks_same_bytes_whoever_traces_them(rdr, scenes):
    """More than 32 samples per pixel, float32 form: a pixel's samples are added up in blocks of 32, the blocks' sums in order
    (include/ptr.h).  The usual path gives every block of a tile to a wave of its own and adds up afterwards
    (render_ext_f32_split_kernel + ext_combine_kernel); the entry point that may not allocate has no room for the blocks' sums
    and keeps them in registers (render_ext_f32_blocks_kernel).  Bytes and float32 colours must be the same either way, for
    whole frames, bands and cyclic strips, with a last block that is not full (100 = 32 + 32 + 32 + 4), with planes too - and
    within the float32 tolerance of the float64 specification."""
    import torch
    from ptr_amd import synthetic as syn
    F = ptr_amd.MODE_F32_ONLY
    stream = torch.cuda.current_stream().cuda_stream
    try:
        for planes in (False, True):
            if planes:
                soa, pl, ph = syn.planes_scene()
                rdr.set_camera(None); rdr.set_scene(soa); rdr.set_background(syn.BACKGROUND)
                rdr.set_extension(pl, ph)
            else:
                _load(rdr, scenes["example1"])
            for (w, h, y0, rows, depth, spp) in ((200, 120, 0, 120, 3, 100), (97, 61, 7, 40, 2, 33), (64, 48, 0, 48, 1, 64)):
                p = rdr.params(w, h, y0, rows, flags=F, max_depth=depth, spp=spp, seed=11)
                a8 = torch.zeros((rows, w), dtype=torch.int32, device="cuda"); af = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
                b8 = torch.zeros_like(a8); bf = torch.zeros_like(af)
                rdr.render_device(p, 0, 1, a8.data_ptr(), af.data_ptr(), stream=stream)
                rdr.sync()
                rdr.render_device_noalloc(p, 0, 1, b8.data_ptr(), bf.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                assert torch.equal(a8, b8) and torch.equal(af, bf), (planes, w, h, spp)
                f64 = rdr.render(w, h, y0, rows, flags=X, max_depth=depth, spp=spp, seed=11)
                got = a8.cpu().numpy().view(np.uint32).ravel()
                assert _psnr(got, f64) >= 45.0, (planes, w, h, spp, _psnr(got, f64))
            # cyclic strips: the parts' rows equal the whole frame's
            w, h, spp = 120, 72, 70
            whole = rdr.render(w, h, flags=F, max_depth=2, spp=spp, seed=3).reshape(h, w)
            for part_no in range(3):
                part = ptr_amd.StripPart(3, part_no, 8)
                rows = part.local_rows(h)
                buf = torch.zeros((rows, w), dtype=torch.int32, device="cuda")
                p = rdr.params(w, h, 0, rows, flags=F, max_depth=2, spp=spp, seed=3)
                rdr.render_strips_device(p, part, 0, 1, buf.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                assert np.array_equal(buf.cpu().numpy().view(np.uint32), whole[part.global_rows(h)]), (planes, part_no)
    finally:
        rdr.set_camera(None)
        rdr.set_extension(None, None)


def test_wave_level_float64_kernel_equals_the_lane_by_lane_one(rdr, scenes):
    """Scenes of spheres take render_ext_f64_kernel (scalar sphere records, tile culling of the camera ray, the shadow loop's
    pair table, square roots and reciprocals by the compiler's own Newton sequences without their range patches); the
    lane-by-lane render_ext_kernel<double> is the specification written out per pixel (PTR_DEBUG_EXT_LANEWISE).  Same
    bytes and the same float32 colours: moved, turned and wide cameras (the culling cone), bands that do not start on a tile
    row, cyclic strips, several samples, scenes whose coordinates are scaled to both ends of what the culls' float32 margins
    were built for."""
    import math
    import torch
    LW = ptr_amd.DEBUG_EXT_LANEWISE
    cams = [None,
            dict(origin=(3.0, 1.5, 4.0)),
            dict(origin=(-6.0, 2.0, -8.0), right=(math.cos(0.6), 0.0, math.sin(0.6)), up=(0.0, 1.0, 0.0),
                 forward=(math.sin(0.6), 0.0, -math.cos(0.6))),
            dict(origin=(0.0, 12.0, -14.0), right=(1.0, 0.0, 0.0), up=(0.0, math.cos(1.0), -math.sin(1.0)),
                 forward=(0.0, -math.sin(1.0), -math.cos(1.0)), tan_half_fov=math.tan(math.radians(50))),
            dict(origin=(0.0, 0.0, -20.0))]                          # the eye inside the transparent sphere
    try:
        for name in ("example1", "random_2", "random_5", "coincident", "inside_sphere"):
            if name not in scenes:
                continue
            _load(rdr, scenes[name])
            for cam in cams:
                if cam is None:
                    rdr.set_camera(None)
                else:
                    rdr.set_camera(**cam)
                for (w, h, y0, rows, depth, spp) in ((333, 201, 0, 201, 5, 1), (256, 160, 13, 90, 3, 3), (97, 61, 0, 61, 7, 2)):
                    a, af = rdr.render(w, h, y0, rows, flags=X, max_depth=depth, spp=spp, seed=9, want_f32=True)
                    b, bf = rdr.render(w, h, y0, rows, flags=X | LW, max_depth=depth, spp=spp, seed=9, want_f32=True)
                    assert np.array_equal(a, b), (name, cam, w, h, depth, spp)
                    assert np.array_equal(af.view(np.uint32), bf.view(np.uint32)), (name, cam, w, h, depth, spp)
        # scaled scenes: coordinates 1e-3 ... 1e4 times the example's
        ex = orc.scene_to_soa(scenes["example1"]["elements"])
        rdr.set_camera(None)
        for k in (1e-3, 0.05, 30.0, 1e4):
            s = ex.copy(); s[0:4] *= k
            rdr.set_scene(s); rdr.set_background([0.7, 1.3, 0.2])
            a = rdr.render(240, 135, flags=X, max_depth=5)
            b = rdr.render(240, 135, flags=X | LW, max_depth=5)
            assert np.array_equal(a, b), k
        # planes and Phong terms (the kernel's GEN form): plane lights, a mirror wall, even / odd / unit exponents; cameras as above
        from ptr_amd import synthetic as syn
        soa, planes, phong = syn.planes_scene()
        for ext in ((planes, phong), (planes, None), (None, phong[:, :soa.shape[1]])):
            rdr.set_scene(soa); rdr.set_background(syn.BACKGROUND); rdr.set_extension(*ext)
            for cam in cams:
                if cam is None:
                    rdr.set_camera(None)
                else:
                    rdr.set_camera(**cam)
                for (w, h, depth, spp) in ((320, 180, 5, 1), (121, 67, 3, 2)):
                    a, af = rdr.render(w, h, flags=X, max_depth=depth, spp=spp, seed=4, want_f32=True)
                    b, bf = rdr.render(w, h, flags=X | LW, max_depth=depth, spp=spp, seed=4, want_f32=True)
                    assert np.array_equal(a, b) and np.array_equal(af.view(np.uint32), bf.view(np.uint32)), (ext[0] is None, ext[1] is None, cam, w, h)
        rdr.set_extension(None, None)
        rdr.set_camera(None)
        # cyclic strips (8 rows: whole tiles, culled; 5 rows: not)
        _load(rdr, scenes["example1"])
        w, h = 200, 120
        whole = rdr.render(w, h, flags=X | LW, max_depth=4).reshape(h, w)
        for strip in (8, 5):
            for part_no in range(3):
                part = ptr_amd.StripPart(3, part_no, strip)
                rows = part.local_rows(h)
                buf = torch.zeros((rows, w), dtype=torch.int32, device="cuda")
                p = rdr.params(w, h, 0, rows, flags=X, max_depth=4)
                rdr.render_strips_device(p, part, 0, 1, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert np.array_equal(buf.cpu().numpy().view(np.uint32), whole[part.global_rows(h)]), (strip, part_no)
    finally:
        rdr.set_camera(None)
        rdr.set_extension(None, None)
