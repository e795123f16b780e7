"""Pins the CPU oracle (oracle/ptr_oracle.c) bit-for-bit to outputs of the unmodified
reference run under Node (tests/golden/, made by oracle/gen_golden.mjs)."""
import hashlib
import os

import numpy as np
import pytest

import oracle_py as orc


def _frame(golden_dir, rel):
    return np.fromfile(os.path.join(golden_dir, rel), dtype=np.uint32)


def test_reference_angle_matches_v8():
    # Math.tan(Math.PI*0.5*30/180) under V8 (RayTracer.js:154); value printed by Node 12
    assert orc.lib().orc_reference_angle() == 0.2679491924311227


def test_small_frames_bit_exact(golden_dir, golden_index, scenes):
    assert len(golden_index["frames"]) >= 20
    for key, meta in golden_index["frames"].items():
        name = key.split("@")[0]
        sc = scenes[name]
        got = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], meta["width"], meta["height"])
        want = _frame(golden_dir, meta["file"])
        assert got.shape == want.shape, key
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"{key}: {bad.size} pixels differ, first at {bad[:5]}"


@pytest.mark.parametrize("key", ["example1@640x480", "example1@1920x1080", "anim_t1234@1920x1080",
                                 "random_5@640x480", "example1@3840x2160"])
def test_big_frame_hashes(golden_index, scenes, key):
    meta = golden_index["hashes"][key]
    sc = scenes[key.split("@")[0]]
    w, h = meta["width"], meta["height"]
    got = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], w, h, threads=8)
    assert hashlib.sha256(got.tobytes()).hexdigest() == meta["sha256"]
    for x, y, v in meta["probes"]:
        assert int(got[y * w + x]) == v
    # F5: last stored row is untouched zeros, pixel (0,0) is clamped-white background
    assert not got[(h - 1) * w:].any()


MORE_HASHES = ["random_0@640x480", "random_1@640x480", "random_2@640x480", "random_3@640x480", "random_4@640x480", "random_6@640x480",
               "random_7@640x480", "inside_sphere@640x480", "odd_colours@640x480", "coincident@640x480", "light_below_horizon@640x480",
               "occluder_behind_light@640x480", "no_lights@320x240", "example1_permuted@800x600", "anim_t98765@1280x720",
               "example1@1001x563", "random_7@1279x719"]


@pytest.mark.parametrize("key", MORE_HASHES)
def test_more_reference_hashes(golden_index, scenes, key):
    """hashes alone of the other golden scenes at larger and odd sizes (the unmodified reference under Node, oracle/gen_golden.mjs)"""
    meta = golden_index["hashes"][key]
    sc = scenes[key.split("@")[0]]
    w, h = meta["width"], meta["height"]
    got = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], w, h, threads=8)
    assert hashlib.sha256(got.tobytes()).hexdigest() == meta["sha256"]
    for x, y, v in meta["probes"]:
        assert int(got[y * w + x]) == v


def test_survey_anchor_hashes(golden_index):
    # SURVEY.md §7 step 1 anchors, measured by the survey session on the unmodified reference
    assert golden_index["hashes"]["example1@640x480"]["sha256"] == \
        "e2ece997c1509ecfa42733bd66d4e45d73ea202c296ecbf62472ec0a470be40e"
    assert golden_index["hashes"]["example1@1920x1080"]["sha256"] == \
        "b2b9c765b7275e3d91f734fa80ff834670448cc4f44fadef02f035b5f88979b0"


def test_banded_render(golden_dir, golden_index, scenes):
    for key, meta in golden_index["bands"].items():
        sc = scenes[key.split("@")[0]]
        soa = orc.scene_to_soa(sc["elements"])
        w, h, jobs, band = meta["width"], meta["height"], meta["jobs"], meta["band"]
        parts = [orc.render(soa, sc["background"], w, h, i * band, band) for i in range(jobs)]
        got = np.concatenate(parts)
        want = _frame(golden_dir, meta["file"])
        assert np.array_equal(got, want), key
        # every band: first row dropped, last stored row zero (F5)
        for p in parts:
            assert not p[(band - 1) * w:].any()


def test_trace_vectors_bit_exact(golden_dir, golden_index, scenes):
    total = 0
    for name, meta in golden_index["trace"].items():
        rec = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.float64).reshape(-1, 9)
        assert rec.shape[0] == meta["count"]
        sc = scenes[name]
        soa = orc.scene_to_soa(sc["elements"])
        for q in rec:
            got = orc.trace(soa, sc["background"], q[0:3], q[3:6])
            # bitwise comparison (also distinguishes -0/+0 and NaN payload-free equality)
            assert got.tobytes() == q[6:9].tobytes(), (name, q)
        total += rec.shape[0]
    assert total >= 1000


def test_request_pixel(golden_dir, golden_index, scenes):
    meta = golden_index["request_pixel"]["example1"]
    rec = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.float64).reshape(-1, 6)
    sc = scenes["example1"]
    soa = orc.scene_to_soa(sc["elements"])
    for q in rec:
        got = orc.request_pixel(soa, sc["background"], q[0:3])
        assert np.array_equal(got, q[3:6])


def test_intersect_table(golden_dir, golden_index):
    meta = golden_index["intersect"]
    rec = np.fromfile(os.path.join(golden_dir, meta["file"]), dtype=np.float64).reshape(-1, 13)
    exits = [0, 0]
    for q in rec:
        hit, t0, t1 = orc.intersect(q[0:3], q[3:6], q[6:9], q[9])
        assert hit == bool(q[10])
        assert np.array([t0, t1]).tobytes() == q[11:13].tobytes()
        exits[int(hit)] += 1
    assert exits[0] > 50 and exits[1] > 50


def test_counters_match_survey(scenes):
    # SURVEY.md §6 "Ray mix @640x480" measured on the instrumented reference
    sc = scenes["example1"]
    _, cnt = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], 640, 480, counters=True)
    assert cnt["primary"] == 307200
    assert cnt["hits"] == 240687
    assert cnt["shadow_rays"] == 481374
    assert cnt["exit_tca"] + cnt["exit_d2"] + cnt["exit_hit"] == 4371301


def test_pack_js_semantics():
    assert orc.pack([2.0, 2.0, 2.0]) == 0xFFFFFFFF
    assert orc.pack([0.0, 0.0, 0.0]) == 0xFF000000
    assert orc.pack([0.5 / 255, 1.5 / 255, 2.5 / 255]) == (0xFF000000 | (3 << 16) | (2 << 8) | 1)
    assert orc.pack([float("nan"), 0.5, 1.0]) == (0xFF000000 | (255 << 16) | (128 << 8) | 0)
    # negative channel: Math.round(-0.2*255) = -51 -> ToInt32 -> OR smears the high bits
    assert orc.pack([-0.2, 0.0, 0.0]) == 0xFFFFFFCD
