"""the render cache's per-sample ray trace (rendercache-js Sample.doRaytracing ->
Raytracer.trace).  tests/golden/rc/ holds what the UNMODIFIED render-cache classes return under Node
(oracle/gen_golden_rc.mjs); the oracle must reproduce it bit for bit, and so must the GPU entry point
ptr_trace_samples_f64 through the C ABI."""
import json
import os

import numpy as np
import pytest

import oracle_py as orc


def _cases(golden_dir):
    with open(os.path.join(golden_dir, "rc", "index.json")) as f:
        index = json.load(f)
    for name, meta in index["samples"].items():
        rec = np.fromfile(os.path.join(golden_dir, "rc", meta["file"])).reshape(-1, 18)
        assert rec.shape[0] == meta["rays"]
        yield name, index["scenes"][name], rec


def _same(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64)) or np.array_equal(a, b)     # -0.0 == 0.0 allowed


def test_oracle_reproduces_the_render_cache_samples(golden_dir):
    n = 0
    for name, sc, rec in _cases(golden_dir):
        soa = orc.scene_to_soa(sc["elements"])
        rgb, rgb8, hit, nrm = orc.trace_samples(soa, sc["background"], rec[:, 0:3], rec[:, 3:6])
        assert _same(rgb, rec[:, 6:9]), name
        assert _same(hit, rec[:, 9:12]), name
        assert _same(nrm, rec[:, 12:15]), name
        assert np.array_equal(rgb8, rec[:, 15:18]), name
        n += rec.shape[0]
    assert n >= 3000


def test_golden_covers_hits_misses_and_inside(golden_dir):
    for name, sc, rec in _cases(golden_dir):
        miss = np.abs(rec[:, 9:12] - (rec[:, 0:3] + rec[:, 3:6] * 1e5)).max(axis=1) < 1e-6
        assert miss.any() and (~miss).any(), name


@pytest.mark.gpu
def test_gpu_samples_bit_exact(golden_dir):
    import ptr_amd
    r = ptr_amd.Renderer(0)
    try:
        for name, sc, rec in _cases(golden_dir):
            r.set_elements(sc["elements"])
            r.set_background(sc["background"])
            for o in np.unique(rec[:, 0:3], axis=0):
                sel = (rec[:, 0:3] == o).all(axis=1)
                rgb, rgb8, hit, nrm = r.trace_samples(o, rec[sel, 3:6])
                assert _same(rgb, rec[sel, 6:9]), name
                assert _same(hit, rec[sel, 9:12]), name
                assert _same(nrm, rec[sel, 12:15]), name
                assert np.array_equal(rgb8.astype(np.float64), rec[sel, 15:18]), name
    finally:
        r.close()
