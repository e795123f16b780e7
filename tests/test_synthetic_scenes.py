"""Synthetic scenes of BASELINE.json configs 3 and 5 (ptr_amd/synthetic.py): the generator is the
definition, so pin it (the reference has no such scenes)."""
import hashlib

import numpy as np
import pytest

import oracle_py as orc
from ptr_amd import synthetic as syn


def test_numpy_philox_equals_the_specification():
    rng = np.random.default_rng(5)
    ctr = rng.integers(0, 2 ** 32, size=(16, 4), dtype=np.uint64).astype(np.uint32)
    got = syn.philox4x32_10(ctr, (0xC0FFEE, 0))
    for row, c in zip(got, ctr):
        assert list(row) == orc.philox([int(v) for v in c], [0xC0FFEE, 0])


def test_example_scene_is_the_reference_list(scenes):
    assert np.array_equal(syn.example1(), orc.scene_to_soa(scenes["example1"]["elements"]))
    c3 = syn.config3_scene()
    assert list(c3[7]) == [0, 0, 1, 0, 1, 0, 0] and not c3[8].any()


def test_many_spheres_is_pinned():
    s = syn.many_spheres()
    assert s.shape == (12, 10000)
    assert hashlib.sha256(s.tobytes()).hexdigest() == MANY_SHA
    small = s[:, 1:-2]
    assert small[0].min() >= -60 and small[0].max() <= 60 and small[1].min() >= -4 and small[1].max() <= 30
    assert small[2].min() >= -120 and small[2].max() <= -10 and small[3].min() >= 0.1 and small[3].max() <= 0.6
    assert (s[9:12].max(axis=0) > 0).sum() == 2
    assert np.array_equal(syn.many_spheres(100)[:, 1:50], s[:, 1:50])     # a prefix of the same stream


MANY_SHA = "06e4854f357095f57de52c9903a939d1d2f928aa7698cb0abcb3991873738e2e"
