"""Soak test (GPU) of the extension modes: the float64 kernel (the default mode) against the CPU specification
(oracle/ptr_oracle.c orc_render_ext) on random scenes with reflective and transparent spheres, random depth, samples and
frame size, a third of them with planes and Phong terms - every byte must be equal - and, beside it, what the float32 form leaves (pixels beyond 1/255, PSNR).
    python tools/soak_ext.py [scenes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), os.path.join(ROOT, "oracle"), ROOT]
import numpy as np
import ptr_amd
import oracle_py as orc                      # checker only



def soak(n_scenes, seed, verbose=True):
    """-> (pixels, pixels where the float64 kernel differs from the CPU specification, pixels where the float32 form is
    beyond 1/255, worst float32 PSNR)"""
    rng = np.random.default_rng(seed)
    r = ptr_amd.Renderer(0)
    bad_total = px_total = beyond_total = 0
    worst_psnr = 1e9
    t0 = time.time()
    for k in range(n_scenes):
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
        bg = rng.uniform(0, 2.0, 3)
        w, h = int(rng.integers(9, 200)), int(rng.integers(5, 140))
        depth, spp, seed = int(rng.integers(0, 8)), int(rng.choice([1, 1, 1, 2, 3, 5])), int(rng.integers(0, 2**31))
        if depth == 0 and spp == 1:
            depth = 1
        soa = np.ascontiguousarray(soa)
        # every third scene: 1-3 planes (a floor, walls, a tilted emissive panel) and / or Phong terms
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
                    planes[9, q] = float(rng.random() < 0.3); planes[10, q] = 0.0
                    if q == 2:
                        planes[11:14, q] = rng.uniform(0.1, 0.5, 3)
            if n_pl == 0 or rng.random() < 0.6:
                ne = n + n_pl
                phong = np.zeros((2, ne)); phong[1] = 1.0
                sel = rng.random(ne) < 0.5
                phong[0, sel] = rng.uniform(0.1, 0.8, int(sel.sum())); phong[1, sel] = rng.integers(1, 200, int(sel.sum()))
        r.set_scene(soa); r.set_background(bg); r.set_extension(planes, phong)
        got = r.render(w, h, flags=0, max_depth=depth, spp=spp, seed=seed)
        if planes is None and phong is None:
            want = orc.render_ext(soa, bg, w, h, max_depth=depth, spp=spp, seed=seed)
        else:
            want = orc.render_ext2(soa, bg, w, h, planes=planes, phong=phong, max_depth=depth, spp=spp, seed=seed)
        bad = int((got != want).sum())
        f32 = r.render(w, h, flags=ptr_amd.MODE_F32_ONLY, max_depth=depth, spp=spp, seed=seed)
        a = f32.view(np.uint8).reshape(-1, 4)[:, :3].astype(np.int32); b = want.view(np.uint8).reshape(-1, 4)[:, :3].astype(np.int32)
        d = np.abs(a - b)
        beyond = int((d.max(axis=1) > 1).sum())
        mse = float((d.astype(np.float64) ** 2).mean())
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        worst_psnr = min(worst_psnr, psnr)
        bad_total += bad; px_total += w * h; beyond_total += beyond
        if verbose and (bad or k % 25 == 0):
            print(f"scene {k}: {n} spheres {w}x{h} depth {depth} spp {spp}: float64 differs on {bad} pixels; float32 beyond 1/255 on {beyond}, "
                  f"PSNR {psnr:.1f} dB   [{time.time() - t0:.0f} s]", flush=True)

    r.close()
    return px_total, bad_total, beyond_total, worst_psnr


if __name__ == "__main__":
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    px_total, bad_total, beyond_total, worst_psnr = soak(n_scenes, int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
    print(f"{n_scenes} scenes, {px_total} pixels: float64 kernel differs from the CPU specification on {bad_total} pixels; "
          f"float32 form beyond 1/255 on {beyond_total} pixels ({100.0 * beyond_total / px_total:.4f} %), worst PSNR {worst_psnr:.1f} dB")
    sys.exit(1 if bad_total else 0)
