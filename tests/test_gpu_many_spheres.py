"""BASELINE.json config 5: 10 000 spheres, brute-force intersection.  The reference semantics
(depth 0) are checked against the oracle, which is pinned by the reference's golden vectors; the
scene itself is synthetic (ptr_amd/synthetic.py), the reference has none of this size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_py as orc          # noqa: E402  (checker only)
import ptr_amd                   # noqa: E402
from ptr_amd import synthetic as syn   # noqa: E402


@pytest.fixture(scope="module")
def rdr():
    r = ptr_amd.Renderer(0)
    r.set_camera(None)
    r.set_scene(syn.many_spheres())
    r.set_background(syn.BACKGROUND)
    yield r
    r.close()


@pytest.fixture(scope="module")
def want():
    return orc.render(syn.many_spheres(), syn.BACKGROUND, 320, 180, flags=0, threads=8)


@pytest.mark.parametrize("mode", ["exact", "hybrid"])
def test_10k_spheres_bytes_equal_the_oracle(rdr, want, mode):
    got = rdr.render(320, 180, flags=ptr_amd.MODE_EXACT_F64 if mode == "exact" else 0)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{mode}: {bad.size} pixels differ, first {bad[:5]}"
    if mode == "hybrid":
        print(f"hybrid 10k spheres: {rdr.stats()['retraced']} of {want.size} pixels re-traced in float64")


def test_10k_spheres_float32_tolerance(rdr, want):
    got = rdr.render(320, 180, flags=ptr_amd.MODE_F32_ONLY)
    d = np.abs(got.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
    off = int((d > 1).sum())
    print(f"f32 10k spheres: {off} of {d.size} pixels beyond 1/255")
    assert off <= 0.01 * d.size                    # 10 000 silhouettes in 57 600 pixels


def test_10k_spheres_band_of_the_full_hd_frame(rdr):
    want = orc.render(syn.many_spheres(), syn.BACKGROUND, 1920, 1080, 500, 6, flags=0, threads=8)
    got = rdr.render(1920, 1080, 500, 6, flags=0)
    assert np.array_equal(got, want)


def test_10k_spheres_depth1_specification(rdr):
    want = orc.render_ext(syn.many_spheres(), syn.BACKGROUND, 160, 90, max_depth=1, spp=1, threads=8)
    got = rdr.render(160, 90, flags=ptr_amd.MODE_EXACT_F64, max_depth=1)
    assert np.array_equal(got, want)


# ---- the many-sphere kernel forced onto the reference's own scenes (golden vectors of the unmodified
# ---- reference): same bytes as every other mode
M = ptr_amd.DEBUG_FORCE_MANY
Q = ptr_amd.QUIRK_ROW_SHIFT


@pytest.fixture(scope="module")
def rdr2():
    r = ptr_amd.Renderer(0)
    yield r
    r.close()


def _load(r, sc):
    r.set_camera(None)
    r.set_elements(sc["elements"])
    r.set_background(sc["background"])


def test_forced_many_kernel_golden_frames(rdr2, golden_dir, golden_index, scenes):
    import os
    for key, meta in golden_index["frames"].items():
        _load(rdr2, scenes[key.split("@")[0]])
        got = rdr2.render(meta["width"], meta["height"], flags=Q | M)
        want = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.uint32)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"{key}: {bad.size} pixels differ, first {bad[:5]}"


@pytest.mark.parametrize("key", ["example1@1920x1080", "example1@3840x2160", "anim_t1234@1920x1080", "random_5@640x480",
                                 "random_7@1279x719", "random_6@640x480", "inside_sphere@640x480", "coincident@640x480",
                                 "example1@1001x563", "anim_t98765@1280x720"])
def test_forced_many_kernel_big_hashes(rdr2, golden_index, scenes, key):
    import hashlib
    meta = golden_index["hashes"][key]
    _load(rdr2, scenes[key.split("@")[0]])
    got = rdr2.render(meta["width"], meta["height"], flags=Q | M)
    assert hashlib.sha256(got.tobytes()).hexdigest() == meta["sha256"]
    print(f"{key}: {rdr2.stats()['retraced']} of {got.size} pixels re-traced")


def test_forced_many_kernel_bands_and_rgb32f(rdr2, scenes):
    sc = scenes["example1"]
    _load(rdr2, sc)
    soa = orc.scene_to_soa(sc["elements"])
    full = orc.render(soa, sc["background"], 333, 211, flags=0).reshape(211, 333)
    parts = [rdr2.render(333, 211, y0, h, flags=M).reshape(h, 333) for y0, h in ((0, 1), (1, 100), (101, 110))]
    assert np.array_equal(np.vstack(parts), full)
    got, f32 = rdr2.render(333, 211, flags=M, want_f32=True)
    ref = orc.render_rgb64(soa, sc["background"], 333, 211)
    err = np.abs(f32[..., :3] - ref).max(axis=2).ravel()
    assert (err > 1e-3).sum() <= 0.01 * err.size               # float32 colour, except on discontinuities


def test_forced_many_kernel_f32_only(rdr2, scenes):
    for name in ("example1", "random_4", "random_7", "inside_sphere"):
        sc = scenes[name]
        _load(rdr2, sc)
        want = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], 320, 240, flags=0)
        got = rdr2.render(320, 240, flags=M | ptr_amd.MODE_F32_ONLY)
        d = np.abs(got.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
        assert (d > 1).sum() <= 0.005 * d.size, name
