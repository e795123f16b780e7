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
