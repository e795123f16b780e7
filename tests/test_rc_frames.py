"""SURVEY.md §8f ranks 1-4 together: whole render-cache frames (rendercache-js Driver.nextFrame) with the cache resident on
the GPU — ptr_rc_cache_* / ptr_rc_next_frame / ptr_rc_get_frames through the C ABI.  tests/golden/rc/frames_* hold what the
UNMODIFIED reference produced under Node over several frames with a moving camera and a seeded Math.random
(oracle/gen_golden_rc_frames.mjs): statistics, colour frame and sampling frame of every frame must come out bit for bit.

The first frame's Driver.initializeCache draws its pixels from Math.random and is the caller's part (include/ptr.h); the
few lines of host logic it needs — the paged allocator and the request loop — are restated below for the test, citing the
reference, with the trace itself going through ptr_trace_samples_f64."""
import json
import os

import numpy as np
import pytest

NEW_PIXEL_DEPTH = 10000000000000.0      # Pixel.js:8


def xorshift32(seed):
    s = (seed & 0xffffffff) or 1
    while True:
        s ^= (s << 13) & 0xffffffff
        s ^= s >> 17
        s ^= (s << 5) & 0xffffffff
        yield s / 4294967296.0


def js_round(x):
    return int(np.floor(x + 0.5))


def dir_to_xy(x, y, w, h, proj_inv, world):
    """Camera.computeDirToXY (rendercache-js/src/Camera.js:39-50) over THREE.Raycaster.setFromCamera, float64 op for op"""
    f = np.float64
    v = [f(x) / f(w) * f(2) - f(1), f(y) / f(h) * f(2) - f(1), f(0.5)]
    for e in (proj_inv, world):
        ww = f(1) / (e[3] * v[0] + e[7] * v[1] + e[11] * v[2] + e[15])
        v = [(e[0] * v[0] + e[4] * v[1] + e[8] * v[2] + e[12]) * ww, (e[1] * v[0] + e[5] * v[1] + e[9] * v[2] + e[13]) * ww,
             (e[2] * v[0] + e[6] * v[1] + e[10] * v[2] + e[14]) * ww]
    v = [v[0] - world[12], v[1] - world[13], v[2] - world[14]]
    ln = np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    s = f(1) / (ln if ln != 0 else f(1))
    v = [c * s for c in v]
    l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2]
    if l2 > 0:
        inv = f(1) / np.sqrt(l2)
        v = [c * inv for c in v]
    return v


class HostCache:
    """Driver.prepare + Driver.initializeCache (rendercache-js/src/Driver.js:81-88, :252-353) on flat arrays"""

    def __init__(self, w, h, ratio, cache_factor=4):
        self.w, self.h, self.st = w, h, w + 2
        cells, n = (w + 2) * (h + 2), int(w * h * cache_factor)
        self.n = n
        self.max_samples = w * h * (1 / ratio)
        self.ratio = ratio
        self.state = {
            "item_pixel": np.full(n, -1, np.int32), "item_hit": np.full(3 * n, np.nan), "item_age": np.zeros(n),
            "item_rgb": np.zeros(n, np.uint32), "pixel_element": np.full(cells, -1, np.int32),
            "pixel_depth": np.full(cells, NEW_PIXEL_DEPTH), "pixel_rgb": np.zeros(cells, np.uint32),
            "pixel_priority": np.zeros(cells, np.int32), "pixel_sample": np.zeros(cells, np.uint8)}
        self.cache_pointer = 0

    def alloc(self, pixel):                                     # allocCacheItem (:252-305)
        s = self.state
        stop = self.cache_pointer + 8
        if stop > self.n:
            stop = 8
        take, oldest, max_age = -1, -1, -1.0
        for i in range(stop - 8, stop):
            if s["item_pixel"][i] < 0:
                take = i
                break
            if s["item_age"][i] > max_age:
                max_age, oldest = s["item_age"][i], i
        if take < 0:                                            # the page is full: Sample.clear() on its oldest item
            take = oldest
            s["item_hit"][3 * take] = np.nan; s["item_age"][take] = -1; s["item_rgb"][take] = 0
        s["item_pixel"][take] = pixel
        s["pixel_element"][pixel] = take
        self.cache_pointer = stop
        return take

    def initialize(self, renderer, rng, cam_from, world, proj_inv, age_factor=1.0):     # initializeCache (:328-353)
        s = self.state
        usage = 0.0
        while usage < 0.2:
            requests = []
            i = 0
            while i < self.max_samples:
                x, y = js_round(next(rng) * (self.w - 1)), js_round(next(rng) * (self.h - 1))
                p = y * self.st + x
                s["pixel_sample"][p] = 1                                # addRequest (:311-326)
                item = s["pixel_element"][p]
                requests.append(item if item >= 0 else self.alloc(p))
                i += 1
            live = [it for it in requests if s["item_pixel"][it] >= 0]  # requestSamples (:842-863)
            dirs = np.array([dir_to_xy(s["item_pixel"][it] % self.st, s["item_pixel"][it] // self.st, self.w, self.h, proj_inv, world)
                             for it in live])
            _, rgb8, hit, _ = renderer.trace_samples(cam_from, dirs)
            for k, it in enumerate(live):
                s["item_rgb"][it] = int(rgb8[k, 0]) | (int(rgb8[k, 1]) << 8) | (int(rgb8[k, 2]) << 16)
                s["item_hit"][3 * it:3 * it + 3] = hit[k]
                s["pixel_rgb"][s["item_pixel"][it]] = s["item_rgb"][it]
            for it in requests:                                         # age(ageFactor, requests) (:936-942)
                s["item_age"][it] += age_factor
            usage += 1 / self.ratio


def frame_cases(golden_dir, method="nextFrame"):
    with open(os.path.join(golden_dir, "rc", "frames_index.json")) as f:
        idx = json.load(f)
    with open(os.path.join(golden_dir, "rc", "index.json")) as f:
        scene = json.load(f)["scenes"]["rc_scene2"]
    for name, m in idx["cases"].items():
        if m.get("method", "nextFrame") != method:                # (Driver.nextFrame1SPP: a host form, pinned through Node)
            continue
        rec = np.fromfile(os.path.join(golden_dir, "rc", m["file"])).reshape(m["frames"], m["doubles_per_frame"])
        frames = np.fromfile(os.path.join(golden_dir, "rc", m["frames_file"]), dtype=np.uint32).reshape(m["frames"], 2, -1)
        yield name, m, scene, rec, frames


def test_golden_frames_are_complete(golden_dir):
    n = 0
    for name, m, scene, rec, frames in frame_cases(golden_dir):
        assert frames.shape[2] == m["width"] * m["height"], name
        assert np.isfinite(rec[:, 68:74]).all(), name
        n += m["frames"]
    assert n == 58
    n = 0
    for name, m, scene, rec, frames in frame_cases(golden_dir, "nextFrame1SPP"):
        assert frames.shape[2] == m["width"] * m["height"], name
        assert np.isfinite(rec[:, 68:74]).all(), name
        n += m["frames"]
    assert n == 13


@pytest.mark.gpu
def test_gpu_resident_frames_bit_exact(golden_dir):
    import ptr_amd
    for name, m, scene, rec, frames in frame_cases(golden_dir):
        r = ptr_amd.Renderer(0)
        try:
            r.set_elements(scene["elements"])
            r.set_background(scene["background"])
            w, h = m["width"], m["height"]
            host = HostCache(w, h, m["ratio"], m["cache_factor"])
            v = rec[0]
            t = m["tunables"]
            host.initialize(r, xorshift32(m["seed"]), v[0:3], v[3:19], v[51:67], t["ageFactor"])
            r.rc_cache_create(w, h, host.n)
            r.rc_cache_upload(host.state, host.cache_pointer)
            for k in range(m["frames"]):
                v = rec[k]
                st = r.rc_next_frame(v[0:3], v[3:19], v[19:35], v[35:51], v[51:67], host.max_samples, t["ageFactor"], t["depthThreshold"],
                                     t["interpolationZero"], t["interpolationRandom"], t["priorityMax"])
                want = v[67:74]
                tag = f"{name} frame {k}"
                assert (np.isnan(st["threshold"]) and np.isnan(want[0])) or st["threshold"] == want[0], tag
                got = [st[key] for key in ("request_factor", "candidates", "requests", "completeness", "total_priority", "cache_usage")]
                assert got == list(want[1:7]), tag
                f = r.rc_get_frames(("color", "sampling"))
                assert np.array_equal(f["color"], frames[k, 0]), tag + " colour frame"
                assert np.array_equal(f["sampling"], frames[k, 1]), tag + " sampling frame"
            # the state comes back whole: a second cache fed with the download continues identically
            state, cp = r.rc_cache_download()
            assert (state["item_pixel"] >= 0).sum() == round(want[6] * host.n)
        finally:
            r.close()


@pytest.mark.gpu
def test_gpu_resident_frames_argument_errors(golden_dir):
    import ptr_amd
    r = ptr_amd.Renderer(0)
    try:
        ident = np.eye(4).reshape(16)
        with pytest.raises(ptr_amd.PtrError, match="no render cache"):
            r.rc_next_frame([0, 0, 0], ident, ident, ident, ident, 10)
        with pytest.raises(ptr_amd.PtrError, match="at least one page"):
            r.rc_cache_create(2, 1, 4)
        r.rc_cache_create(8, 4, 128)
        with pytest.raises(ptr_amd.PtrError, match="no state yet"):
            r.rc_next_frame([0, 0, 0], ident, ident, ident, ident, 10)
        host = HostCache(8, 4, 4)
        bad = dict(host.state)
        bad["pixel_element"] = host.state["pixel_element"].copy()
        bad["pixel_element"][5] = 128
        with pytest.raises(ptr_amd.PtrError, match="not a cache item"):
            r.rc_cache_upload(bad, 0)
        with pytest.raises(ptr_amd.PtrError, match="multiple of 8"):
            r.rc_cache_upload(host.state, 12)
        r.rc_cache_upload(host.state, 0)
        with pytest.raises(ptr_amd.PtrError, match="no scene"):
            r.rc_next_frame([0, 0, 0], ident, ident, ident, ident, 10)
        r.set_scene(np.zeros((12, 0)))
        with pytest.raises(ptr_amd.PtrError, match="priority_max"):
            r.rc_next_frame([0, 0, 0], ident, ident, ident, ident, 10, pmax=300.0)
        # an empty cache and an empty scene: every pixel is a gap at priority 255, a quarter of them is requested
        st = r.rc_next_frame([0, 0, 0], ident, ident, ident, ident, 8)
        assert st["candidates"] == 32 and st["requests"] == 8 and st["threshold"] == 255
        r.rc_cache_destroy()
    finally:
        r.close()


def pick_loop(rf, n_cand):
    """Driver.directSamples' pick loop (rendercache-js/src/Driver.js:795-822), statement for statement"""
    out, index, last = [], 0.0, None
    while index < n_cand:
        i = int(np.floor(index + 0.5))                          # Math.round of a non-negative number
        if last is None or last != i:
            if i < n_cand:
                out.append(i)
        last = i
        index += rf
    return np.array(out, dtype=np.int32)


def test_pick_sequence_closed_form_equals_the_running_sum():
    """the library finds the picks piece by piece of the float64 running sum (host code, no GPU): same numbers as adding
    it up, for request factors with exact ties (dyadic, halves), just above 1, below 1, larger than the candidate count"""
    import ptr_amd
    rng = np.random.default_rng(5)
    cases = [(1.0, 1), (1.0, 1000), (0.999999, 777), (0.25, 9), (1.5, 2), (2.5, 100001), (3.0, 3), (1e9, 12), (7.123456789, 1),
             (1.0000000000000002, 5000), (16.000000000000004, 250000), (1.1 + 2.2, 99999), (255.5, 2073600), (4.000000001, 2073600)]
    for t in range(500):
        c = int(rng.integers(1, 200000)) if t % 3 else int(rng.integers(1, 40))
        kind = t % 5
        rf = (float(rng.uniform(1, 40)), float(rng.integers(1, 1 << 20)) / float(1 << int(rng.integers(0, 20))),
              float(rng.uniform(0.05, 1.0)), 1.0 + float(rng.uniform(0, 1e-3)), float(int(rng.integers(2, 60))) + 0.5)[kind]
        if c / rf > 5e4:
            rf = c / 5e4 + 1
        cases.append((rf, c))
    for rf, c in cases:
        assert np.array_equal(ptr_amd.rc_pick_sequence(rf, c), pick_loop(rf, c)), (rf, c)
    assert ptr_amd.rc_pick_sequence(3.0, 0).size == 0
    with pytest.raises(ptr_amd.PtrError):
        ptr_amd.rc_pick_sequence(0.0, 10)                       # the reference's loop would not end
