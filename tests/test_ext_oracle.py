"""Extension modes (max_depth > 0, spp > 1) — CPU side.  The reference has neither,
so nothing here is reference parity: the float64 specification in oracle/ptr_oracle.c is checked
against published Philox known answers, against the reference semantics where the two must agree
(max_depth 0, spp 1), against structural properties, and against its own frozen vectors
(tests/golden/ext, made by oracle/gen_golden_ext.py: "self-pinned, parity unpinned")."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_py as orc


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("name", ["example1", "random_2", "inside_sphere", "empty"])
def test_depth0_spp1_is_the_reference_path(scenes, name):
    sc = scenes[name]
    soa = orc.scene_to_soa(sc["elements"])
    a = orc.render(soa, sc["background"], 96, 64, flags=0)
    b = orc.render_ext(soa, sc["background"], 96, 64, max_depth=0, spp=1)
    assert np.array_equal(a, b)


def test_scene_without_mirrors_ignores_depth(scenes):
    sc = scenes["occluder_behind_light"]
    soa = orc.scene_to_soa(sc["elements"])
    a = orc.render_ext(soa, sc["background"], 96, 64, max_depth=0)
    b = orc.render_ext(soa, sc["background"], 96, 64, max_depth=5)
    assert np.array_equal(a, b)


def test_bands_and_thread_counts_agree(scenes):
    sc = scenes["example1"]
    soa = orc.scene_to_soa(sc["elements"])
    full = orc.render_ext(soa, sc["background"], 80, 60, max_depth=3, spp=3, seed=5, threads=1).reshape(60, 80)
    top = orc.render_ext(soa, sc["background"], 80, 60, 0, 17, max_depth=3, spp=3, seed=5, threads=3).reshape(17, 80)
    bot = orc.render_ext(soa, sc["background"], 80, 60, 17, 43, max_depth=3, spp=3, seed=5, threads=8).reshape(43, 80)
    assert np.array_equal(full, np.vstack([top, bot]))
    other = orc.render_ext(soa, sc["background"], 80, 60, max_depth=3, spp=3, seed=6).reshape(60, 80)
    assert not np.array_equal(full, other)


def test_planes_and_phong_properties():
    """the plane / Phong part of the specification on its own terms: a floor plane where the reference scene has its
    radius-10000 sphere gives nearly the same picture; ks = 0 is the plain extension mode; a plane behind the light does
    not darken the scene (planes block in front of the light only); the exponent is an exact product of squares"""
    from ptr_amd import synthetic as syn
    ex = syn.example1()
    floor = np.zeros((14, 1)); floor[1, 0] = -4.0; floor[4, 0] = 1.0; floor[6:9, 0] = 0.2
    a = orc.render_ext(ex, syn.BACKGROUND, 96, 64, max_depth=0).view(np.uint8).astype(int)
    b = orc.render_ext2(ex[:, 1:], syn.BACKGROUND, 96, 64, planes=floor, max_depth=0).view(np.uint8).astype(int)
    assert (np.abs(a - b) > 2).mean() < 0.02          # (the horizon line and shadow edges move by a pixel)
    zero = np.zeros((2, 7)); zero[1] = 5
    assert np.array_equal(orc.render_ext2(ex, syn.BACKGROUND, 96, 64, phong=zero, max_depth=2),
                          orc.render_ext(ex, syn.BACKGROUND, 96, 64, max_depth=2))
    wall = np.zeros((14, 1)); wall[2, 0] = -100.0; wall[5, 0] = 1.0; wall[6:9, 0] = 0.5       # far behind everything, lights included
    c = orc.render_ext2(ex, syn.BACKGROUND, 96, 64, planes=wall, max_depth=0).view(np.uint8).astype(int).reshape(64, 96, 4)
    assert np.array_equal(c[40:], a.reshape(64, 96, 4)[40:])            # the lit floor and spheres are untouched
    shiny = np.zeros((2, 7)); shiny[0, 1] = 1.0; shiny[1] = 1; shiny[1, 1] = 64
    d = orc.render_ext2(ex, syn.BACKGROUND, 96, 64, phong=shiny, max_depth=0).view(np.uint8).astype(int)
    assert (d >= a).all() and (d > a).sum() > 20                         # a highlight only ever adds light


def test_frozen_vectors(golden_dir, scenes):
    with open(os.path.join(golden_dir, "ext", "index.json")) as f:
        index = json.load(f)
    for key, c in index["cases"].items():
        if c["width"] * c["height"] * c["spp"] > 640 * 480 * 4:
            continue                                             # the 1080p case: GPU test only
        if c["scene"] == "synthetic.planes_scene":               # planes + Phong terms (extension data)
            from ptr_amd import synthetic as syn
            soa, planes, phong = syn.planes_scene()
            out = orc.render_ext2(soa, syn.BACKGROUND, c["width"], c["height"], planes=planes, phong=phong,
                                  max_depth=c["max_depth"], spp=c["spp"], seed=c["seed"])
        else:
            sc = scenes[c["scene"]]
            out = orc.render_ext(orc.scene_to_soa(sc["elements"]), sc["background"], c["width"], c["height"],
                                 max_depth=c["max_depth"], spp=c["spp"], seed=c["seed"])
        assert hashlib.sha256(out.tobytes()).hexdigest() == c["sha256"], key
        if "file" in c:
            want = np.fromfile(os.path.join(golden_dir, c["file"]), dtype=np.uint32)
            assert np.array_equal(out, want), key
