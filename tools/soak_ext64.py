"""Soak test (GPU): the wave-level float64 bounce-loop kernel (render_ext_f64_kernel: scalar records, tile culling of the camera
ray, pair-table culling of the shadow loop, mid-range sqrt / reciprocal sequences) against the lane-by-lane kernel
(PTR_DEBUG_EXT_LANEWISE, itself pinned to the CPU specification by tests/ and tools/soak_ext.py) on many random scenes of
spheres, planes and Phong terms: bytes AND float32 colours must be identical.  Scenes: the generator of tools/soak_ext.py,
plus scaled coordinates (1e-4 ... 1e5), huge and tiny radii, lights inside spheres, moved / turned / wide cameras, bands.
Usage: python tools/soak_ext64.py [scenes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np
import ptr_amd
if os.environ.get("PTR_AB_LIB"):
    ptr_amd.LIB_PATH = os.path.abspath(os.environ["PTR_AB_LIB"])

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 64064)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # render this scene alone and list its differing pixels
X, LW = ptr_amd.MODE_EXACT_F64, ptr_amd.DEBUG_EXT_LANEWISE
r = ptr_amd.Renderer(0)
px_total = bad_total = 0
t0 = time.time()
for k in range(n_scenes):
    n = int(rng.integers(1, 13)) if k % 10 else int(rng.integers(13, 80))
    soa = np.zeros((12, n))
    soa[0] = rng.uniform(-9, 9, n); soa[1] = rng.uniform(-3, 6, n); soa[2] = rng.uniform(-45, -6, n)
    soa[3] = rng.uniform(0.2, 4.5, n)
    if k % 11 == 0:
        soa[3] *= rng.choice([1e-3, 0.05, 8.0], n)                 # tiny and big spheres among the others
    soa[4:7] = rng.uniform(0.0, 1.0, (3, n))
    soa[7] = (rng.random(n) < 0.6) * rng.uniform(0.1, 1.0, n)
    soa[8] = (rng.random(n) < 0.4) * rng.uniform(0.1, 1.0, n)
    lights = rng.random(n) < 0.25
    lights[rng.integers(0, n)] = True
    soa[9:12, lights] = rng.uniform(0.1, 2.5, (3, int(lights.sum())))
    if k % 3 == 0:
        soa[:, 0] = [0, -10000 - rng.uniform(2, 6), -20, 10000, 0.2, 0.2, 0.2, float(k % 6 == 0), 0, 0, 0, 0]
    if k % 7 == 0 and n > 2:                                       # a sphere inside another, two that overlap
        soa[0:3, 1] = soa[0:3, 2]; soa[3, 1] = 0.5 * soa[3, 2]
    if k % 13 == 0 and n > 1:                                      # a light inside a (transparent) sphere
        j = int(np.flatnonzero(lights)[0]); i = (j + 1) % n
        soa[0:3, i] = soa[0:3, j]; soa[3, i] = 3.0 * max(soa[3, j], 0.1); soa[8, i] = 0.7
    if k % 5 == 4:
        soa[0:4] *= float(rng.choice([1e-4, 1e-2, 50.0, 1e3, 1e5]))   # the whole scene scaled
    if k % 4 == 1:
        org = rng.uniform(-8, 8, 3); fwd = rng.normal(size=3); fwd /= np.linalg.norm(fwd)
        up0 = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.95 else np.array([1.0, 0.0, 0.0])
        right = np.cross(fwd, up0); right /= np.linalg.norm(right)
        r.set_camera(org, right, np.cross(right, fwd), fwd, tan_half_fov=float(rng.uniform(0.1, 1.5)))
    else:
        r.set_camera(None)
    # every third scene: 1-3 planes (a floor, walls, a tilted emissive panel) and / or Phong terms (the kernel's GEN form)
    planes = phong = None
    if k % 3 == 1:
        n_pl = int(rng.integers(0, 4))
        if n_pl:
            planes = np.zeros((14, n_pl))
            for q in range(n_pl):
                nrm = rng.normal(size=3) if q else np.array([0.0, 1.0, 0.0])
                nrm /= np.sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2])
                pt = np.array([0.0, -4.0 - rng.uniform(0, 3), -20.0]) if q == 0 else rng.uniform(-40, 40, 3) + np.array([0, 0, -30.0])
                planes[0:3, q] = pt; planes[3:6, q] = nrm
                planes[6:9, q] = rng.uniform(0.1, 0.9, 3)
                planes[9, q] = float(rng.random() < 0.3); planes[10, q] = float(rng.random() < 0.2) * 0.5
                if q == 2 or rng.random() < 0.2:
                    planes[11:14, q] = rng.uniform(0.1, 0.5, 3)
        if n_pl == 0 or rng.random() < 0.6:
            ne = n + n_pl
            phong = np.zeros((2, ne)); phong[1] = 1.0
            sel = rng.random(ne) < 0.5
            phong[0, sel] = rng.uniform(0.1, 0.8, int(sel.sum())); phong[1, sel] = rng.integers(1, 200, int(sel.sum()))
    r.set_scene(np.ascontiguousarray(soa)); r.set_background(rng.uniform(0, 2.0, 3)); r.set_extension(planes, phong)
    w, h = int(rng.integers(9, 400)), int(rng.integers(5, 260))
    y0 = int(rng.integers(0, h)) if k % 6 == 5 else 0
    rows = int(rng.integers(1, h - y0 + 1)) if k % 6 == 5 else h
    depth, spp, seed = int(rng.integers(0, 9)), int(rng.choice([1, 1, 1, 2, 3])), int(rng.integers(0, 2**31))
    if depth == 0 and spp == 1:
        depth = 1
    if only >= 0 and k != only:
        continue
    a, af = r.render(w, h, y0, rows, flags=X, max_depth=depth, spp=spp, seed=seed, want_f32=True)
    b, bf = r.render(w, h, y0, rows, flags=X | LW, max_depth=depth, spp=spp, seed=seed, want_f32=True)
    bad = int((a != b).sum()) + int((af.view(np.uint32) != bf.view(np.uint32)).any(axis=-1).sum())
    px_total += w * rows; bad_total += bad
    if only >= 0:
        np.set_printoptions(precision=17, linewidth=200)
        print("spheres (rows x y z r sr sg sb refl transp er eg eb):"); print(soa.T)
        print("planes:"); print(None if planes is None else planes.T); print("phong:"); print(None if phong is None else phong.T)
        idx = np.flatnonzero((a != b) | (af.view(np.uint32) != bf.view(np.uint32)).any(axis=-1).reshape(-1))
        for i in idx[:12]:
            print("pixel", int(i % w), int(i // w) + y0, hex(int(a[i])), hex(int(b[i])), af.reshape(-1, 4)[i], bf.reshape(-1, 4)[i])
        for dd in range(0, depth + 1):
            a2 = r.render(w, h, y0, rows, flags=X, max_depth=dd, spp=spp, seed=seed)
            b2 = r.render(w, h, y0, rows, flags=X | LW, max_depth=dd, spp=spp, seed=seed)
            print("depth", dd, "differ:", int((a2 != b2).sum()))
    if bad or k % 500 == 0:
        print(f"scene {k}: {n} spheres, {0 if planes is None else planes.shape[1]} planes"
              f"{'' if planes is None else ' (transparent: %d, emissive: %d)' % (int((planes[10] > 0).sum()), int((planes[11:14].max(axis=0) > 0).sum()))}"
              f"{', Phong terms' if phong is not None else ''}, {w}x{rows} depth {depth} spp {spp}: {bad} pixels differ   [{time.time() - t0:.0f} s]", flush=True)
r.set_camera(None)
print(f"{n_scenes} scenes, {px_total} pixels: the wave-level float64 kernel differs from the lane-by-lane one on {bad_total} pixels")
sys.exit(1 if bad_total else 0)
