"""The hybrid mode against the CPU oracle on scenes built to strain its float32 error bounds (the soak
tool compares with the GPU's own float64 kernel, and only at everyday scales).  Bounded: a few small frames per case.
Reference sites the cases aim at: Sphere.intersect's two exits (RT/src/Sphere.js:18-25), the t0 < 0 switch and the far
clip (RT/src/RayTracer.js:32-37), the any-hit shadow test and its 1e-4 bias (:74-85)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_py as orc          # noqa: E402  (checker only)
import ptr_amd                   # noqa: E402
from ptr_amd import synthetic as syn   # noqa: E402

W, H = 192, 108


def _oracle_general(soa, bg, w, h, org, right, up, fwd, t):
    R, U, F = (np.asarray(v, dtype=np.float64) for v in (right, up, fwd))
    xs = (2 * ((np.arange(w) + 0.5) * (1 / w)) - 1) * t * (w / h)
    ys = (1 - 2 * ((np.arange(h) + 0.5) * (1 / h))) * t
    xx, yy = np.meshgrid(xs, ys)
    d = [xx * R[k] + yy * U[k] + F[k] for k in range(3)]
    inv = 1 / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    dirs = np.stack([d[k] * inv for k in range(3)], axis=-1).reshape(-1, 3)
    rgb = orc.trace_batch(soa, np.asarray(bg, dtype=np.float64), np.tile(np.asarray(org, dtype=np.float64), (w * h, 1)), dirs)[0]
    return np.array([orc.pack(v) for v in rgb], dtype=np.uint32)


def _cases():
    ex = syn.example1()
    out = {}
    for k in (1e3, 3e4, 1e-3, 1e-2):                    # the whole scene scaled about the camera: coordinates 1e-2 .. 1e8
        s = ex.copy(); s[0:4] *= k
        out[f"scaled x{k:g}"] = s
    s = ex.copy(); s[3, 0] = 6.0e8; s[1, 0] = -6.0e8 - 4.0        # floor so large that the own-sphere proof is switched off
    out["floor r=6e8 (self_skip off)"] = s
    s = ex.copy(); s[0:4, 2] = [0.02, 0.01, -0.3, 1e-3]            # a 1 mm sphere 30 cm from the eye
    s[0:4, 3] = [3e5, 2e5, -1.2e6, 1e6]                            # and a 1000 km one
    out["r=1e-3 and r=1e6"] = s
    s = ex.copy(); s[0:4, 5] = [0.0, -4.0, -20.0, 0.0]             # a zero-radius light whose centre lies ON the floor
    out["zero-radius light on a hit point"] = s
    s = ex.copy(); s[0:4, 5] = [0.0, 0.0, -16.0, 0.25]             # a light centred on the red sphere's nearest point
    out["light centred on a surface point"] = s
    s = ex.copy(); s[3, 4] = 0.0; s[0:3, 3] = s[0:3, 1]; s[3, 3] = 4.0   # zero-radius element, two coincident spheres
    out["zero radius + coincident"] = s
    s = ex.copy(); s[2, 1:5] += 19.9                               # spheres around and right in front of the eye
    s[0:4, 1] = [0.0, 0.0, 0.5, 4.0]                               # the eye inside the red one, 0.5 from its centre
    out["eye inside a sphere"] = s
    s = ex.copy(); s[0:3, 6] = [0.0, 1e-3, -24.0 + 1e-3]           # a light a millimetre outside the red sphere's far side
    out["light grazing a sphere"] = s
    return out


@pytest.mark.parametrize("name", list(_cases().keys()))
def test_hybrid_equals_the_oracle_reference_camera(name):
    soa = _cases()[name]
    bg = np.array([0.7, 1.3, 0.2])
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(soa); r.set_background(bg)
        for (w, h) in ((W, H), (97, 61)):
            want = orc.render(soa, bg, w, h, flags=0)
            got = r.render(w, h, flags=0)
            bad = np.flatnonzero(got != want)
            assert bad.size == 0, f"{name} {w}x{h}: {bad.size} pixels differ from the oracle, first {[(int(i % w), int(i // w), hex(int(got[i])), hex(int(want[i]))) for i in bad[:4]]}"
            exact = r.render(w, h, flags=ptr_amd.MODE_EXACT_F64)
            assert np.array_equal(exact, want), name
    finally:
        r.close()


@pytest.mark.parametrize("org", [(1.0e4, 5.0e3, -2.0e4), (-3.0e5, 7.0, 1.0e5), (0.3, 1.0e-4, 2.0e-3)])
def test_hybrid_equals_the_oracle_camera_far_from_the_origin(org):
    """the scene moved along with the camera: the float64 reference loses digits that the camera-relative float32 records
    do not have in the first place; the bytes must still be the reference's"""
    ex = syn.example1()
    soa = ex.copy(); soa[0:3] += np.asarray(org)[:, None]
    bg = np.array([2.0, 2.0, 2.0])
    fwd = np.array([0.05, -0.08, -1.0]); fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    t = math.tan(math.pi * 0.5 * 30 / 180)
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(soa); r.set_background(bg)
        r.set_camera(org, right, up, fwd, tan_half_fov=t)
        w, h = 128, 72
        want = _oracle_general(soa, bg, w, h, org, right, up, fwd, t)
        got = r.render(w, h, flags=0)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, f"camera at {org}: {bad.size} pixels differ, first {[(int(i % w), int(i // w), hex(int(got[i])), hex(int(want[i]))) for i in bad[:4]]}"
    finally:
        r.close()
