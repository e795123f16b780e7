"""Soak test (GPU): hybrid mode against the float64 kernel on many random few-sphere scenes, cameras and frame sizes.
The float64 kernel is pinned to the CPU oracle and the reference's golden frames by tests/; this script looks for pixels
where an error bound of the float32 kernel is too tight (any difference is a bug).  Usage: python tools/soak_hybrid.py [scenes] [seed] [max_spheres = 12]
(more than 12 spheres: the two-level kernel of ptr_kernels_many.hip)
PTR_SOAK_SCALE=1: every fourth scene has tiny / huge coordinates or radii.  PTR_SOAK_MODE=tolerant: PTR_MODE_TOLERANT instead, and "differ" means "differ by more than one unit in some channel"."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np
import ptr_amd

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
TOL = os.environ.get("PTR_SOAK_MODE") == "tolerant"
one_unit = 0
r = ptr_amd.Renderer(0)
bad_total = px_total = redo_total = 0
for t in range(n_scenes):
    n = int(rng.integers(1, max_n + 1))
    kind = t % 5
    soa = np.zeros((12, n))
    spread = (12, 3, 40)[kind % 3]
    soa[0] = rng.uniform(-spread, spread, n); soa[1] = rng.uniform(-spread / 2, spread / 2, n)
    soa[2] = rng.uniform(-45, -6, n) if kind != 4 else rng.uniform(-30, 30, n)
    soa[3] = rng.uniform(0.05, 5.0, n) * (10.0 if kind == 3 else 1.0)
    soa[4:7] = rng.uniform(0.0, 1.0, (3, n))
    lights = rng.random(n) < (0.25 if kind != 2 else 0.9)
    lights[rng.integers(0, n)] = True
    soa[9:12, lights] = rng.uniform(0.05, 2.0, (3, int(lights.sum())))
    if kind == 0 and n > 1:
        soa[0:4, 0] = [0, -10000 - rng.uniform(2, 6), -20, 10000]; soa[9:12, 0] = 0
    if os.environ.get("PTR_SOAK_SCALE") and t % 4 == 3:            # the whole scene scaled (tiny / huge coordinates), or single spheres
        if t % 8 == 3:
            soa[0:4] *= float(rng.choice([1e-4, 1e-2, 50.0, 1e3, 1e5]))
        else:
            soa[3] *= rng.choice([1e-3, 0.05, 1.0, 8.0], n)
    bg = rng.uniform(0, 2.0, 3)
    if t % 3 == 2:
        org = rng.uniform(-8, 8, 3); fwd = rng.normal(size=3); fwd /= np.linalg.norm(fwd)
        up0 = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.95 else np.array([1.0, 0.0, 0.0])
        right = np.cross(fwd, up0); right /= np.linalg.norm(right)
        r.set_camera(org, right, np.cross(right, fwd), fwd)
    else:
        r.set_camera(None)
    r.set_scene(soa); r.set_background(bg)
    w, h = int(rng.integers(17, 700)), int(rng.integers(9, 400))
    got = r.render(w, h, flags=ptr_amd.MODE_TOLERANT if TOL else 0)
    st = r.stats()
    want = r.render(w, h, flags=ptr_amd.MODE_EXACT_F64)
    if TOL:
        d = np.abs(got.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
        bad = int((d > 1).sum()); one_unit += int((d == 1).sum())
    else:
        bad = int((got != want).sum())
    bad_total += bad; px_total += w * h; redo_total += st["retraced"] + st["reshaded"]
    if bad:
        idx = np.flatnonzero(got != want)[:3]
        print(f"scene {t} (kind {kind}, n={n}, {w}x{h}): {bad} pixels differ, first {[(int(i % w), int(i // w), hex(int(got[i])), hex(int(want[i]))) for i in idx]}", flush=True)
    if (t + 1) % 20000 == 0 and t + 1 < n_scenes:            # a long run must show that it is alive
        print(f"... {t + 1} scenes, {px_total} pixels, {bad_total} differ so far", flush=True)
print(f"{n_scenes} scenes, {px_total} pixels, {redo_total} re-done in float64 ({100.0 * redo_total / px_total:.2f} %), {bad_total} differ"
      + (f" by more than one unit; {one_unit} by one unit ({100.0 * one_unit / px_total:.4f} %)" if TOL else ""))
sys.exit(1 if bad_total else 0)
