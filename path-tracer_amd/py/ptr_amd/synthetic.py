"""Synthetic scenes for the BASELINE.json configurations that the reference has no scene for.

The reference ships one scene list (RT/examples/example1.js:9-45); configs 3 and 5 of BASELINE.json
ask for variations of it.  Both are generated, not stored: the generator is the definition.
Planes are the C-ABI order (include/ptr.h): cx cy cz radius sr sg sb reflection transparency er eg eb.
"""
import numpy as np

EXAMPLE1 = np.array([
    # cx     cy       cz   radius   sr    sg    sb   refl transp  er   eg   eb      RT/examples/example1.js:9-45
    [0.0, -10004.0, -20.0, 10000.0, 0.20, 0.20, 0.20, 0.0, 0.0, 0.0, 0.0, 0.0],
    [0.0, 0.0, -20.0, 4.0, 1.00, 0.32, 0.36, 1.0, 0.5, 0.0, 0.0, 0.0],
    [5.0, -1.0, -15.0, 2.0, 0.90, 0.76, 0.46, 1.0, 0.0, 0.0, 0.0, 0.0],
    [5.0, 0.0, -25.0, 3.0, 0.65, 0.77, 0.97, 1.0, 0.0, 0.0, 0.0, 0.0],
    [-5.5, 0.0, -15.0, 3.0, 0.90, 0.90, 0.90, 1.0, 0.0, 0.0, 0.0, 0.0],
    [0.0, 20.0, -30.0, 3.0, 0.00, 0.00, 0.00, 0.0, 0.0, 1.2, 1.2, 1.2],
    [0.0, 10.0, 10.0, 3.0, 0.00, 0.00, 0.00, 0.0, 0.0, 1.0, 1.0, 1.0],
], dtype=np.float64)
BACKGROUND = (2.0, 2.0, 2.0)


def philox4x32_10(ctr, key):
    """Vectorised Philox4x32-10.  ctr: (m, 4) uint32, key: (2,) -> (m, 4) uint32."""
    c = [np.ascontiguousarray(ctr[:, k], dtype=np.uint64) for k in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & m32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, axis=1).astype(np.uint32)


def example1():
    """(12, 7) planes of the reference's example scene."""
    return np.ascontiguousarray(EXAMPLE1.T)


def config3_scene():
    """Config 3: the example scene with the four coloured spheres alternating diffuse / reflective."""
    s = EXAMPLE1.copy()
    for row, refl in zip((1, 2, 3, 4), (0.0, 1.0, 0.0, 1.0)):
        s[row, 7] = refl
        s[row, 8] = 0.0
    return np.ascontiguousarray(s.T)


def many_spheres(n=10000, key=0xC0FFEE):
    """Config 5: floor + 2 lights of the example scene + (n - 3) small spheres, centres uniform in
    x [-60, 60], y [-4, 30], z [-120, -10], radius [0.1, 0.6], colour [0.1, 1]^3; sphere i draws its
    seven numbers from Philox4x32-10(counter (i, 0|1, 0, 0), key (key, 0)), u = (word + 0.5) / 2^32.
    Scene order: floor, the small spheres, the two lights."""
    m = n - 3
    idx = np.arange(m, dtype=np.uint32)
    z = np.zeros(m, dtype=np.uint32)
    a = philox4x32_10(np.stack([idx, z, z, z], axis=1), (key, 0))
    b = philox4x32_10(np.stack([idx, z + 1, z, z], axis=1), (key, 0))
    u = (np.concatenate([a, b], axis=1).astype(np.float64) + 0.5) / 4294967296.0
    soa = np.zeros((12, n), dtype=np.float64)
    soa[:, 0] = EXAMPLE1[0]
    soa[0, 1:1 + m] = -60.0 + 120.0 * u[:, 0]
    soa[1, 1:1 + m] = -4.0 + 34.0 * u[:, 1]
    soa[2, 1:1 + m] = -120.0 + 110.0 * u[:, 2]
    soa[3, 1:1 + m] = 0.1 + 0.5 * u[:, 3]
    soa[4, 1:1 + m] = 0.1 + 0.9 * u[:, 4]
    soa[5, 1:1 + m] = 0.1 + 0.9 * u[:, 5]
    soa[6, 1:1 + m] = 0.1 + 0.9 * u[:, 6]
    soa[7, 1:1 + m] = (u[:, 7] < 0.25).astype(np.float64)       # a quarter of them mirrors (depth >= 1 only)
    soa[:, n - 2] = EXAMPLE1[5]
    soa[:, n - 1] = EXAMPLE1[6]
    return np.ascontiguousarray(soa)


def planes_scene():
    """Planes + Phong terms (extension data, include/ptr.h): the example scene's coloured spheres and lights over a floor
    PLANE y = -4 (instead of the radius-10000 sphere), a mirror wall z = -45, a tilted emissive panel; Phong highlights on
    the spheres (even, odd and unit exponents) and on the floor.  Returns (soa (12, 6), planes (14, 3), phong (2, 9))."""
    soa = np.ascontiguousarray(EXAMPLE1[1:].T)                  # without the floor sphere: 4 spheres + 2 lights
    # the panel's normal as Plane's constructor makes it from (-0.6, -0.8, 0): Vector3.normalize, invLen = 1 / sqrt(len2)
    raw = (-0.6, -0.8, 0.0)
    inv = 1.0 / np.sqrt(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2])
    pn = [raw[0] * inv, raw[1] * inv, raw[2] * inv]
    planes = np.array([
        # px   py    pz     nx   ny        nz        sr    sg    sb   refl transp  er   eg   eb
        [0.0, -4.0, 0.0,   0.0, 1.0,      0.0,      0.2,  0.2,  0.2, 0.0, 0.0,   0.0, 0.0, 0.0],
        [0.0, 0.0, -45.0,  0.0, 0.0,      1.0,      0.8,  0.8,  0.9, 1.0, 0.0,   0.0, 0.0, 0.0],
        [30.0, 25.0, -20.0, pn[0], pn[1], pn[2],    0.0,  0.0,  0.0, 0.0, 0.0,   0.3, 0.25, 0.2],
    ]).T
    n = soa.shape[1] + planes.shape[1]
    phong = np.zeros((2, n))
    phong[1] = 1.0
    phong[0, 0:4] = [0.6, 0.4, 0.5, 0.3]; phong[1, 0:4] = [32, 37, 1, 128]
    phong[0, 6] = 0.25; phong[1, 6] = 8                         # the floor plane
    return soa, np.ascontiguousarray(planes), phong
