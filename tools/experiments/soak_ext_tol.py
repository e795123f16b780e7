"""Soak test (GPU) of PTR_MODE_TOLERANT in the extension modes: the checked float32 bounce loop (+ float64 re-do of the pixels it
doubts) against the float64 kernel (the specification, pinned to the CPU oracle by tests/ and tools/soak_ext.py) on random
scenes of reflective / transparent / nested spheres, random depth, samples and frame size.  The bar is north_star's: no
channel of any pixel further than 1 from the specification's byte.  Beside it: what the unchecked float32 form leaves, and
the share of pixels re-done.
    python tools/soak_ext_tol.py [scenes] [seed] [max frame edge]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np
import ptr_amd


def scene(rng, k):
    n = int(rng.integers(1, 13)) if k % 10 else int(rng.integers(13, 40))
    soa = np.zeros((12, n))
    soa[0] = rng.uniform(-9, 9, n); soa[1] = rng.uniform(-3, 6, n); soa[2] = rng.uniform(-45, -6, n)
    soa[3] = rng.uniform(0.2, 4.5, n)
    soa[4:7] = rng.uniform(0.0, 1.0, (3, n))
    soa[7] = (rng.random(n) < 0.6) * rng.uniform(0.1, 1.0, n)
    soa[8] = (rng.random(n) < 0.4) * rng.uniform(0.1, 1.0, n)
    lights = rng.random(n) < 0.25
    lights[rng.integers(0, n)] = True
    soa[9:12, lights] = rng.uniform(0.1, 2.5, (3, int(lights.sum())))
    if k % 3 == 0:
        soa[:, 0] = [0, -10000 - rng.uniform(2, 6), -20, 10000, 0.2, 0.2, 0.2, float(k % 6 == 0), 0, 0, 0, 0]
    if k % 7 == 0 and n > 2:                                  # a sphere inside another, two that overlap
        soa[0:3, 1] = soa[0:3, 2]; soa[3, 1] = 0.5 * soa[3, 2]
    return np.ascontiguousarray(soa), rng.uniform(0, 2.0, 3)


def beyond(a, b):
    d = np.abs(a.view(np.uint8).reshape(-1, 4).astype(np.int16) - b.view(np.uint8).reshape(-1, 4).astype(np.int16)).max(axis=1)
    return int((d > 1).sum()), int(d.max())


def soak(n_scenes, seed, edge=400, verbose=True):
    rng = np.random.default_rng(seed)
    r = ptr_amd.Renderer(0)
    px = bad_tol = bad_f32 = redo = 0
    worst = 0
    t0 = time.time()
    for k in range(n_scenes):
        soa, bg = scene(rng, k)
        w, h = int(rng.integers(9, edge)), int(rng.integers(5, edge))
        depth, spp, sd = int(rng.integers(0, 8)), int(rng.choice([1, 1, 1, 2, 3, 5])), int(rng.integers(0, 2**31))
        if depth == 0 and spp == 1:
            depth = 1
        if k % 4 == 3:                                            # arbitrary cameras too
            org = rng.uniform(-6, 6, 3); fwd = rng.normal(size=3); fwd[2] = -abs(fwd[2]) - 0.5; fwd /= np.linalg.norm(fwd)
            right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
            r.set_camera(org, right, np.cross(right, fwd), fwd)
        else:
            r.set_camera(None)
        r.set_scene(soa); r.set_background(bg)
        want = r.render(w, h, flags=0, max_depth=depth, spp=spp, seed=sd)
        got = r.render(w, h, flags=ptr_amd.MODE_TOLERANT, max_depth=depth, spp=spp, seed=sd)
        redo += r.stats()["retraced"]
        f32 = r.render(w, h, flags=ptr_amd.MODE_F32_ONLY, max_depth=depth, spp=spp, seed=sd)
        bt, wt = beyond(got, want); bf, _ = beyond(f32, want)
        px += w * h; bad_tol += bt; bad_f32 += bf; worst = max(worst, wt)
        if verbose and (bt or k % 50 == 0):
            print(f"scene {k}: {soa.shape[1]} spheres {w}x{h} depth {depth} spp {spp}: tolerant beyond 1/255 on {bt} px (worst {wt}), float32 on {bf}; "
                  f"re-done so far {100.0 * redo / max(1, px):.2f} %   [{time.time() - t0:.0f} s]", flush=True)
    r.close()
    return px, bad_tol, bad_f32, redo, worst


if __name__ == "__main__":
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    px, bad_tol, bad_f32, redo, worst = soak(n_scenes, int(sys.argv[2]) if len(sys.argv) > 2 else 2028,
                                             int(sys.argv[3]) if len(sys.argv) > 3 else 400)
    print(f"{n_scenes} scenes, {px} pixels: tolerant mode beyond 1/255 on {bad_tol} pixels (worst channel difference {worst}); "
          f"unchecked float32 on {bad_f32} ({100.0 * bad_f32 / px:.4f} %); re-done in float64: {redo} ({100.0 * redo / px:.2f} %)")
    sys.exit(1 if bad_tol else 0)
