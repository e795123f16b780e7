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


sys.path.insert(0, ROOT)
from bench import algorithmic_flops, algorithmic_flops_ext, PEAK_FP32_VALU_TFLOPS  # noqa: E402
PEAK_FP64_VALU_TFLOPS = PEAK_FP32_VALU_TFLOPS / 2


def work(r, soa, w, h, depth, spp):
    """algorithmic FLOPs and rays of one frame, from the float64 counting kernel (the reference's control flow; the same
    counters as the CPU oracle's, tests/test_gpu_many_spheres.py / test_gpu_ext.py).  Many samples: counted on a frame of
    1/16 of the pixels and scaled."""
    scale = 16 if spp > 4 else 1
    cw, ch = (w // 4, h // 4) if scale == 16 else (w, h)
    tmp = torch.empty(cw * ch, dtype=torch.int32, device="cuda")
    r.render_device(r.params(cw, ch, flags=ptr_amd.FLAG_STATS, max_depth=depth, spp=spp, seed=0x5EED1234), 0, 1, tmp.data_ptr())
    r.sync()
    st = r.stats()
    n_lights = int(((soa[9] > 0) | (soa[10] > 0) | (soa[11] > 0)).sum())
    if depth == 0 and spp == 1:
        return scale * algorithmic_flops(st), scale * (st["primary"] + st["shadow_rays"])
    return scale * algorithmic_flops_ext(st, n_lights), scale * (st["primary"] + st["secondary"] + st["shadow_rays"])


def run(r, name, soa, w, h, flags, depth, spp, reps=3, frames=1):
    r.set_scene(soa)
    r.set_background(syn.BACKGROUND)
    flops, rays = work(r, soa, w, h, depth, spp)
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
    f64 = bool(flags & ptr_amd.MODE_EXACT_F64) or (flags == 0 and (depth or spp > 1))
    peak = PEAK_FP64_VALU_TFLOPS if f64 else PEAK_FP32_VALU_TFLOPS
    ach = flops / (best * 1e-3) / 1e12
    print(json.dumps(dict(case=name, width=w, height=h, max_depth=depth, spp=spp, flags=flags, kernel_ms=round(best, 4),
                          wall_ms=round(wall, 3), Mpaths_per_s=round(w * h * spp / best / 1e3, 1),
                          Mrays_per_s=round(rays / best / 1e3, 1),
                          roofline=dict(bound="valu", achieved=round(ach, 3), peak=round(peak, 1), unit="TFLOP/s",
                                        frac=round(ach / peak, 4), algorithmic_flops=int(flops),
                                        note="algorithmic FLOPs of the reference's / specification's control flow, counted by the "
                                             "float64 counting kernel; hybrid time includes the float64 re-trace"))), flush=True)


def main():
    which = sys.argv[1:] or ["c2", "c3", "c5"]
    r = ptr_amd.Renderer(0)
    X, F = ptr_amd.MODE_EXACT_F64, ptr_amd.MODE_F32_ONLY
    if "c2" in which:
        run(r, "C2 example1 depth0 hybrid", syn.example1(), 1920, 1080, 0, 0, 1)
        run(r, "C2 example1 depth0 tolerant", syn.example1(), 1920, 1080, ptr_amd.MODE_TOLERANT, 0, 1)
        run(r, "C2 example1 depth5 f64", syn.example1(), 1920, 1080, X, 5, 1)
        run(r, "C2 example1 depth5 f32", syn.example1(), 1920, 1080, F, 5, 1)
    if "c3" in which:
        run(r, "C3 64spp depth5 f32", syn.config3_scene(), 1920, 1080, F, 5, 64)
        run(r, "C3 64spp depth5 f64", syn.config3_scene(), 1920, 1080, X, 5, 64, reps=1)
        run(r, "C4-shape 4K 256spp depth5 f32 (one GPU's 1/8 of the rows)", syn.config3_scene(), 3840, 2160 // 8, F, 5, 256, reps=1)
    if "c5" in which:
        big = syn.many_spheres()
        run(r, "C5 10k depth0 hybrid", big, 1920, 1080, 0, 0, 1)
        run(r, "C5 10k depth0 tolerant", big, 1920, 1080, ptr_amd.MODE_TOLERANT, 0, 1)
        run(r, "C5 10k depth0 f32", big, 1920, 1080, F, 0, 1)
        run(r, "C5 10k depth0 f64", big, 1920, 1080, X, 0, 1, reps=1)
        run(r, "C5 10k depth1 f32", big, 1920, 1080, F, 1, 1, reps=1)


if __name__ == "__main__":
    main()
