"""PTR_MODE_TOLERANT (include/ptr.h): the hybrid mode held to north_star's bar - max per-channel error <= 1/255 on EVERY
pixel - instead of to identical bytes.  Checked against (a) the reference's golden frames (bytes of the unmodified
reference under Node), (b) the float64 kernel at 640x480 / 1080p / 4K (itself pinned to the reference's sha256 hashes by
tests/test_gpu_parity.py), (c) the CPU oracle on the adversarial scales of tests/test_gpu_adversarial.py and (d) a
bounded soak over random scenes, cameras and ragged frame sizes.  What the mode queues for the float64 re-trace is the
discontinuous doubts only (RayTracer.js:26-41 hit/miss and nearest, :36 far clip, :74-85 shadow any-hit), so it must re-do
far fewer pixels than the default."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_py as orc          # noqa: E402  (checker only)
import ptr_amd                   # noqa: E402
from ptr_amd import synthetic as syn   # noqa: E402

Q = ptr_amd.QUIRK_ROW_SHIFT
TOL = ptr_amd.MODE_TOLERANT


@pytest.fixture(scope="module")
def rdr():
    r = ptr_amd.Renderer(0)
    yield r
    r.close()


def _load(r, sc):
    r.set_camera(None)
    r.set_elements(sc["elements"])
    r.set_background(sc["background"])


def _max_channel_diff(a, b):
    a8 = a.view(np.uint8).astype(np.int16)
    b8 = b.view(np.uint8).astype(np.int16)
    return int(np.abs(a8 - b8).max()) if a8.size else 0


def test_golden_frames_within_one_unit(rdr, golden_dir, golden_index, scenes):
    for key, meta in golden_index["frames"].items():
        _load(rdr, scenes[key.split("@")[0]])
        got = rdr.render(meta["width"], meta["height"], flags=Q | TOL)
        want = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.uint32)
        assert _max_channel_diff(got, want) <= 1, key          # tolerance: 1/255 per channel, every pixel
        # alpha and the quirk's untouched last row are not colour: they must be the reference's
        assert np.array_equal(got >> 24, want >> 24), key


def test_banded_render_within_one_unit(rdr, golden_dir, golden_index, scenes):
    for key, meta in golden_index["bands"].items():
        _load(rdr, scenes[key.split("@")[0]])
        w, h, jobs, band = meta["width"], meta["height"], meta["jobs"], meta["band"]
        got = np.concatenate([rdr.render(w, h, i * band, band, flags=Q | TOL) for i in range(jobs)])
        want = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.uint32)
        assert _max_channel_diff(got, want) <= 1 and np.array_equal(got >> 24, want >> 24), key


@pytest.mark.parametrize("key", ["example1@640x480", "example1@1920x1080", "example1@3840x2160",
                                 "anim_t1234@1920x1080", "random_5@640x480"])
def test_big_frames_against_the_float64_kernel(rdr, golden_index, scenes, key):
    meta = golden_index["hashes"][key]
    _load(rdr, scenes[key.split("@")[0]])
    w, h = meta["width"], meta["height"]
    exact = rdr.render(w, h, flags=Q | ptr_amd.MODE_EXACT_F64)
    hyb = rdr.render(w, h, flags=Q)
    st_h = rdr.stats()
    got = rdr.render(w, h, flags=Q | TOL)
    st = rdr.stats()
    assert np.array_equal(hyb, exact)
    assert _max_channel_diff(got, exact) <= 1, key             # tolerance: 1/255 per channel, every pixel
    if len(scenes[key.split("@")[0]]["elements"]) <= 12:
        assert st["reshaded"] == 0                             # the mode has no shading-only re-do
        assert 0 < st["retraced"] < 0.5 * (st_h["retraced"] + st_h["reshaded"]), (st, st_h)
    else:                                                      # the many-sphere kernel: its rounding doubts are gone, the others stay
        assert st["retraced"] == st_h["retraced"] and st["reshaded"] < st_h["reshaded"], (st, st_h)
    differ = int((got != exact).sum())
    print(f"{key}: tolerant re-traced {st['retraced']} + {st['reshaded']} px (default {st_h['retraced']} + {st_h['reshaded']}), "
          f"{differ} of {w * h} px differ from the reference by one unit")


def test_more_reference_scenes_within_one_unit(rdr, golden_index, scenes):
    """every other frame whose hash the reference left (tests/test_gpu_parity.py pins the float64 kernel to it): larger and odd sizes"""
    known = ("example1@640x480", "example1@1920x1080", "example1@3840x2160", "anim_t1234@1920x1080", "random_5@640x480")
    seen = 0
    for key, meta in golden_index["hashes"].items():
        if key in known:
            continue
        _load(rdr, scenes[key.split("@")[0]])
        w, h = meta["width"], meta["height"]
        exact = rdr.render(w, h, flags=Q | ptr_amd.MODE_EXACT_F64)
        got = rdr.render(w, h, flags=Q | TOL)
        assert _max_channel_diff(got, exact) <= 1, key             # tolerance: 1/255 per channel, every pixel
        assert np.array_equal(got >> 24, exact >> 24), key
        seen += 1
    assert seen == 17


def test_unshifted_and_float32_outputs(rdr, scenes):
    sc = scenes["example1"]
    _load(rdr, sc)
    soa = orc.scene_to_soa(sc["elements"])
    got = rdr.render(200, 150, 10, 100, flags=TOL)
    want = orc.render(soa, sc["background"], 200, 150, 10, 100, flags=0)
    assert _max_channel_diff(got, want) <= 1


def _adversarial():
    ex = syn.example1()
    out = {}
    for k in (1e3, 3e4, 1e-3, 1e-2):
        s = ex.copy(); s[0:4] *= k
        out[f"scaled x{k:g}"] = s
    s = ex.copy(); s[3, 0] = 6.0e8; s[1, 0] = -6.0e8 - 4.0
    out["floor r=6e8"] = s
    s = ex.copy(); s[0:4, 2] = [0.02, 0.01, -0.3, 1e-3]; s[0:4, 3] = [3e5, 2e5, -1.2e6, 1e6]
    out["r=1e-3 and r=1e6"] = s
    s = ex.copy(); s[0:4, 5] = [0.0, -4.0, -20.0, 0.0]
    out["zero-radius light on a hit point"] = s
    s = ex.copy(); s[0:4, 5] = [0.0, 0.0, -16.0, 0.25]
    out["light centred on a surface point"] = s
    s = ex.copy(); s[2, 1:5] += 19.9; s[0:4, 1] = [0.0, 0.0, 0.5, 4.0]
    out["eye inside a sphere"] = s
    s = ex.copy(); s[0:3, 6] = [0.0, 1e-3, -24.0 + 1e-3]
    out["light grazing a sphere"] = s
    s = ex.copy(); s[9:12, 5] = 400.0; s[9:12, 6] = 250.0          # lights so bright that tol_thr is below any float32 error
    out["very bright lights"] = s
    s = ex.copy(); s[9:12, 5] = 3e-3; s[9:12, 6] = 2e-3; s[4:7] *= 0.5   # a dim scene: every channel below a few units
    out["very dim lights"] = s
    return out


@pytest.mark.parametrize("name", list(_adversarial().keys()))
def test_adversarial_scales_against_the_oracle(name):
    soa = _adversarial()[name]
    bg = np.array([0.7, 1.3, 0.2])
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(soa); r.set_background(bg)
        for w, h in ((192, 108), (61, 47)):
            got = r.render(w, h, flags=TOL)
            want = orc.render(soa, bg, w, h, flags=0, threads=8)
            assert _max_channel_diff(got, want) <= 1, (name, w, h)
    finally:
        r.close()


def test_bounded_soak_against_the_float64_kernel(rdr):
    """240 random few-sphere scenes (the generator of tools/soak_hybrid.py), reference and arbitrary cameras, ragged sizes."""
    rng = np.random.default_rng(31337)
    px = redo = differ = 0
    for t in range(240):
        n = int(rng.integers(1, 13))
        kind = t % 5
        soa = np.zeros((12, n))
        spread = (12, 3, 40)[kind % 3]
        soa[0] = rng.uniform(-spread, spread, n); soa[1] = rng.uniform(-spread / 2, spread / 2, n)
        soa[2] = rng.uniform(-45, -6, n) if kind != 4 else rng.uniform(-30, 30, n)
        soa[3] = rng.uniform(0.05, 5.0, n) * (10.0 if kind == 3 else 1.0)
        soa[4:7] = rng.uniform(0.0, 1.0, (3, n))
        lights = rng.random(n) < (0.25 if kind != 2 else 0.9)
        lights[rng.integers(0, n)] = True
        soa[9:12, lights] = rng.uniform(0.05, 2.0, (3, int(lights.sum())))
        if kind == 0 and n > 1:
            soa[0:4, 0] = [0, -10000 - rng.uniform(2, 6), -20, 10000]; soa[9:12, 0] = 0
        if t % 3 == 2:
            org = rng.uniform(-8, 8, 3); fwd = rng.normal(size=3); fwd /= np.linalg.norm(fwd)
            up0 = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.95 else np.array([1.0, 0.0, 0.0])
            right = np.cross(fwd, up0); right /= np.linalg.norm(right)
            rdr.set_camera(org, right, np.cross(right, fwd), fwd)
        else:
            rdr.set_camera(None)
        rdr.set_scene(soa); rdr.set_background(rng.uniform(0, 2.0, 3))
        w, h = int(rng.integers(17, 500)), int(rng.integers(9, 300))
        got = rdr.render(w, h, flags=TOL)
        redo += rdr.stats()["retraced"]
        want = rdr.render(w, h, flags=ptr_amd.MODE_EXACT_F64)
        assert _max_channel_diff(got, want) <= 1, f"scene {t} (kind {kind}, n={n}, {w}x{h})"
        px += w * h; differ += int((got != want).sum())
    rdr.set_camera(None)
    print(f"soak: {px} px, {redo} re-traced ({100.0 * redo / px:.3f} %), {differ} differ by one unit ({100.0 * differ / px:.3f} %)")
