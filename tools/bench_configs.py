"""Times the BASELINE.json configurations other than the bench line (which bench.py owns):
C2 depth 5, C3 64 spp, C5 10k spheres.  Prints one JSON line per case.  GPU only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "py"))
import torch  # noqa: E402
import ptr_amd  # noqa: E402
from ptr_amd import synthetic as syn  # noqa: E402


def run(r, name, soa, w, h, flags, depth, spp, reps=3, frames=1):
    r.set_scene(soa)
    r.set_background(syn.BACKGROUND)
    out = torch.empty(frames * w * h, dtype=torch.int32, device="cuda")
    p = r.params(w, h, flags=flags | ptr_amd.FLAG_TIMING, max_depth=depth, spp=spp, seed=0x5EED1234)
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r.render_device(p, 0, 1, out.data_ptr())
        r.sync()
        wall = (time.perf_counter() - t) * 1e3
        k0, k1 = r.kernel_ms()
        best = min(best, k0 + k1)
    print(json.dumps(dict(case=name, width=w, height=h, max_depth=depth, spp=spp, flags=flags, kernel_ms=round(best, 4),
                          wall_ms=round(wall, 3), Mpaths_per_s=round(w * h * spp / best / 1e3, 1))), flush=True)


def main():
    which = sys.argv[1:] or ["c2", "c3", "c5"]
    r = ptr_amd.Renderer(0)
    X, F = ptr_amd.MODE_EXACT_F64, ptr_amd.MODE_F32_ONLY
    if "c2" in which:
        run(r, "C2 example1 depth0 hybrid", syn.example1(), 1920, 1080, 0, 0, 1)
        run(r, "C2 example1 depth5 f64", syn.example1(), 1920, 1080, X, 5, 1)
        run(r, "C2 example1 depth5 f32", syn.example1(), 1920, 1080, F, 5, 1)
    if "c3" in which:
        run(r, "C3 64spp depth5 f32", syn.config3_scene(), 1920, 1080, F, 5, 64)
        run(r, "C3 64spp depth5 f64", syn.config3_scene(), 1920, 1080, X, 5, 64, reps=1)
        run(r, "C4-shape 4K 256spp depth5 f32 (one GPU's 1/8 of the rows)", syn.config3_scene(), 3840, 2160 // 8, F, 5, 256, reps=1)
    if "c5" in which:
        big = syn.many_spheres()
        run(r, "C5 10k depth0 hybrid", big, 1920, 1080, 0, 0, 1)
        run(r, "C5 10k depth0 f32", big, 1920, 1080, F, 0, 1)
        run(r, "C5 10k depth0 f64", big, 1920, 1080, X, 0, 1, reps=1)
        run(r, "C5 10k depth1 f32", big, 1920, 1080, F, 1, 1, reps=1)


if __name__ == "__main__":
    main()
