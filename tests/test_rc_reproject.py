"""the render cache's reprojection pass (Driver.reprojectFrame + Camera.reprojectPixel over
three.js Vector3.project).  tests/golden/rc/reproject_*.f64 were produced by the UNMODIFIED reference under Node
with the three.js it vendors (oracle/gen_golden_rc_reproject.mjs); the oracle, and the GPU entry point
ptr_rc_reproject through the C ABI, must reproduce every output bit for bit."""
import json
import os

import numpy as np
import pytest

import oracle_py as orc


def load_cases(golden_dir):
    with open(os.path.join(golden_dir, "rc", "reproject_index.json")) as f:
        index = json.load(f)
    for name, m in index["cases"].items():
        rec = np.fromfile(os.path.join(golden_dir, "rc", m["file"]))
        w, h, n = m["width"], m["height"], m["items"]
        cells = (w + 2) * (h + 2)
        assert rec.size == 35 + 4 * n + 2 * n + 2 * cells
        o = 35
        items = rec[o:o + 4 * n].reshape(n, 4); o += 4 * n
        out_items = rec[o:o + 2 * n].reshape(n, 2); o += 2 * n
        pix = rec[o:].reshape(cells, 2)
        yield name, dict(m, cam_from=rec[0:3], view=rec[3:19], proj=rec[19:35], hits=items[:, 0:3].copy(), age_in=items[:, 3].copy(),
                         age_out=out_items[:, 0], item_pixel=out_items[:, 1].astype(np.int32),
                         pixel_element=pix[:, 0].astype(np.int32), pixel_depth=pix[:, 1])


def check(name, c, age, item_pixel, el, dep):
    w, h = c["width"], c["height"]
    inner = np.zeros((h + 2, w + 2), dtype=bool); inner[1:h + 1, 1:w + 1] = True
    inner = inner.ravel()
    assert np.array_equal(age, c["age_out"]), name
    assert np.array_equal(item_pixel, c["item_pixel"]), name
    assert np.array_equal(el[inner], c["pixel_element"][inner]), name
    assert np.array_equal(dep[inner], c["pixel_depth"][inner]), name


def test_oracle_reproduces_the_reference_pass(golden_dir):
    seen = 0
    for name, c in load_cases(golden_dir):
        out = orc.rc_reproject(c["cam_from"], c["view"], c["proj"], c["width"], c["height"], c["hits"], c["age_in"],
                               c["age_factor"], c["depth_threshold"])
        check(name, c, *out)
        # the fixture exercises every branch: no hit, depth 0 (+4), outside (+2), loser (+1), displaced winner, winner
        d = c["age_out"] - c["age_in"]
        assert set(np.unique(d)) >= ({0.0, 1.0, 4.0} if name.startswith("same") else {0.0, 1.0, 2.0, 4.0}), name
        displaced = (c["item_pixel"] >= 0) & (c["pixel_element"][np.maximum(c["item_pixel"], 0)] != np.arange(c["items"]))
        assert displaced.any(), name
        seen += 1
    assert seen == 3


@pytest.mark.gpu
def test_gpu_reprojection_bit_exact(golden_dir):
    import ptr_amd
    r = ptr_amd.Renderer(0)
    try:
        for name, c in load_cases(golden_dir):
            out = r.rc_reproject(c["cam_from"], c["view"], c["proj"], c["width"], c["height"], c["hits"], c["age_in"],
                                 c["age_factor"], c["depth_threshold"])
            check(name, c, *out)
    finally:
        r.close()


@pytest.mark.gpu
def test_gpu_reprojection_full_hd_against_oracle():
    """1080p cache (4 items per pixel, 8.3 M items) of synthetic hits: the GPU pass equals the sequential oracle."""
    import ptr_amd
    w, h = 1920, 1080
    n = w * h * 4
    rng = np.random.default_rng(7)
    hits = np.stack([rng.uniform(-30, 30, n), rng.uniform(-6, 14, n), rng.uniform(-60, -8, n)], axis=1)
    hits[rng.random(n) < 0.2, 0] = np.nan
    hits[::1000] = hits[1::1000]                       # equal depths: first index must win
    age = rng.integers(0, 30, n).astype(np.float64)
    name, c = next(load_cases(os.path.join(os.path.dirname(__file__), "golden")))
    proj = c["proj"].copy(); proj[0] = c["proj"][5] / (w / h)       # same fov, 16:9
    want = orc.rc_reproject(c["cam_from"], c["view"], proj, w, h, hits, age)
    r = ptr_amd.Renderer(0)
    try:
        got = r.rc_reproject(c["cam_from"], c["view"], proj, w, h, hits, age)
    finally:
        r.close()
    inner = np.zeros((h + 2, w + 2), dtype=bool); inner[1:h + 1, 1:w + 1] = True
    inner = inner.ravel()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2][inner], want[2][inner]) and np.array_equal(got[3][inner], want[3][inner])
    assert (want[1] >= 0).sum() > 100000
