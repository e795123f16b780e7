"""the render cache's depth culling, gap filling and output frames (Driver.depthCulling,
Driver.fillGaps, getColorFrame / getReprojectionFrame / getPriorityFrame).  tests/golden/rc/shade_* were produced by the
UNMODIFIED reference under Node right after its reprojectFrame (oracle/gen_golden_rc_reproject.mjs)."""
import json
import os

import numpy as np
import pytest

import oracle_py as orc
from test_rc_reproject import load_cases


def shade_cases(golden_dir):
    for name, c in load_cases(golden_dir):
        w, h, n = c["width"], c["height"], c["items"]
        cells = (w + 2) * (h + 2)
        rec = np.fromfile(os.path.join(golden_dir, "rc", c["shade"]))
        assert rec.size == 3 * n + n + 4 * cells + 768 + 3
        o = 0
        rgb = rec[o:o + 3 * n].reshape(n, 3); o += 3 * n
        age = rec[o:o + n]; o += n
        pix = rec[o:o + 4 * cells].reshape(cells, 4); o += 4 * cells
        hist = rec[o:o + 768].reshape(3, 256); o += 768
        frames = np.fromfile(os.path.join(golden_dir, "rc", c["shade_frames"]), dtype=np.uint32).reshape(3, w * h)
        yield name, dict(c, item_rgb=rgb, shade_age=age, pixel_rgb=pix[:, 0:3], pixel_priority=pix[:, 3], hist=hist,
                         stats=rec[o:o + 3], frames=frames)


def check_shade(name, c, out):
    w, h = c["width"], c["height"]
    inner = np.zeros((h + 2, w + 2), dtype=bool); inner[1:h + 1, 1:w + 1] = True
    inner = inner.ravel()
    assert np.array_equal(out["age"], c["shade_age"]), name
    assert np.array_equal(out["pixel_rgb"][inner], c["pixel_rgb"][inner]), name
    assert np.array_equal(out["pixel_priority"][inner], c["pixel_priority"][inner]), name
    assert np.array_equal(np.nan_to_num(out["hist"]), np.nan_to_num(c["hist"])), name
    assert np.array_equal(out["stats"], c["stats"]), name
    for k, key in enumerate(("color", "reprojection", "priority")):
        assert np.array_equal(out[key], c["frames"][k]), (name, key)


def test_oracle_reproduces_culling_filling_and_frames(golden_dir):
    for name, c in shade_cases(golden_dir):
        rgb = np.nan_to_num(c["item_rgb"])                         # items never sampled have no colour and no pixel
        out = orc.rc_shade(c["width"], c["height"], c["pixel_element"], c["pixel_depth"], c["age_out"], rgb,
                           c["age_factor"], c["interpolation_zero"], c["interpolation_random"], c["priority_max"])
        check_shade(name, c, out)
        assert (c["shade_age"] - c["age_out"] == 10 * c["age_factor"]).any(), name   # some depth discontinuity was aged
        assert c["hist"][0].sum() > 0 and c["hist"][2].sum() > 0, name


@pytest.mark.gpu
def test_gpu_culling_filling_and_frames_bit_exact(golden_dir):
    import ptr_amd
    r = ptr_amd.Renderer(0)
    try:
        for name, c in shade_cases(golden_dir):
            out = r.rc_shade(c["width"], c["height"], c["pixel_element"], c["pixel_depth"], c["age_out"], np.nan_to_num(c["item_rgb"]),
                             c["age_factor"], c["interpolation_zero"], c["interpolation_random"], c["priority_max"])
            check_shade(name, c, out)
    finally:
        r.close()


@pytest.mark.gpu
def test_gpu_elements_on_border_cells_match_the_cpu_restatement():
    """Driver.initializeCache draws pixel positions from 0, so border cells of the grid can carry an element, and nothing
    ever detaches them: depthCulling and fillGaps then see them as neighbours (their depth stays 1e13, their colour is the one
    they were sampled with).  The whole-frame goldens (tests/golden/rc/frames_*, through Node) depend on this; here the GPU
    pass is compared with the CPU restatement on random grids with many such cells."""
    import ptr_amd
    rng = np.random.default_rng(8804)
    r = ptr_amd.Renderer(0)
    try:
        for w, h in ((37, 21), (64, 48), (5, 3)):
            cells = (w + 2) * (h + 2)
            n = 4 * w * h
            el = np.full(cells, -1, dtype=np.int32)
            grid = el.reshape(h + 2, w + 2)
            take = rng.random((h + 2, w + 2)) < 0.3
            take[0, :] = rng.random(w + 2) < 0.5; take[-1, :] = rng.random(w + 2) < 0.5
            take[:, 0] = rng.random(h + 2) < 0.5; take[:, -1] = rng.random(h + 2) < 0.5
            grid[take] = rng.choice(n, size=int(take.sum()), replace=False)
            dep = np.full(cells, 1e6); dep[el >= 0] = rng.uniform(5, 60, int((el >= 0).sum()))
            border = np.ones((h + 2, w + 2), dtype=bool); border[1:-1, 1:-1] = False
            dep[border.ravel()] = 10000000000000.0                 # Pixel.js:8, never changed on the border
            age = rng.integers(0, 60, n).astype(np.float64)
            rgb = rng.integers(0, 256, (n, 3)).astype(np.float64)
            prgb = np.zeros((cells, 3)); prgb[border.ravel()] = rng.integers(0, 256, (int(border.sum()), 3))
            want = orc.rc_shade(w, h, el, dep, age, rgb, pixel_rgb_in=prgb)
            got = r.rc_shade(w, h, el, dep, age, rgb, pixel_rgb_in=prgb)
            inner = ~border.ravel()
            assert np.array_equal(got["age"], want["age"])
            assert np.array_equal(got["pixel_rgb"][inner], want["pixel_rgb"][inner])
            assert np.array_equal(got["pixel_priority"][inner], want["pixel_priority"][inner])
            assert np.array_equal(got["hist"], want["hist"]) and np.array_equal(got["stats"], want["stats"])
            for key in ("color", "reprojection", "priority"):
                assert np.array_equal(got[key], want[key]), key
            # the border really mattered: masking its elements changes the result
            el2 = el.copy(); el2[border.ravel()] = -1
            assert not np.array_equal(orc.rc_shade(w, h, el2, dep, age, rgb)["pixel_priority"], want["pixel_priority"])
    finally:
        r.close()
