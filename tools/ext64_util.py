"""Lane utilisation of the wave-level float64 bounce loop, from a measurement build of the library
(bash tools/build_variant.sh util "-DPTR_EXT64_UTIL"): each pixel's value is (its ray steps | its wave's << 8 | its steps that
bounce << 16 | its wave's << 24).   PTR_AB_LIB=path-tracer_amd/lib/libptr_hip_util.so python tools/ext64_util.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), ROOT]
import numpy as np, torch, ptr_amd
ptr_amd.LIB_PATH = os.path.abspath(os.environ["PTR_AB_LIB"])
from bench import example1_soa

r = ptr_amd.Renderer(0)
r.set_scene(np.ascontiguousarray(example1_soa()[None])); r.set_background([2.0, 2.0, 2.0])
for depth in (1, 2, 3, 5):      # (depth 0 is the reference-semantics kernel's, not this loop's)
    w, h = 1920, 1080
    out = torch.zeros((1, h, w), dtype=torch.int32, device="cuda")
    p = r.params(w, h, flags=ptr_amd.MODE_EXACT_F64, max_depth=depth, spp=1)
    r.render_device(p, 0, 1, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype(np.uint32)[0]
    lane, wave, bl, bw = v & 255, (v >> 8) & 255, (v >> 16) & 255, v >> 24
    # the wave's passes weigh on all 64 lanes; passes of the last rows' partial tiles count as they are
    print(f"depth {depth}: ray steps per pixel {lane.mean():.3f}, per wave pass-lanes {wave.mean():.3f} -> lanes with a ray "
          f"{lane.sum() / wave.sum():.3f}; bouncing steps {bl.mean():.3f} of {bw.mean():.3f} -> {bl.sum() / max(bw.sum(), 1):.3f}; "
          f"most steps of a pixel {lane.max()}", flush=True)
    t = wave.reshape(h // 8, 8, w // 8, 8)[:, 0, :, 0]
    print("   passes per wave: mean %.2f, share of waves with > 2 passes %.3f, their share of all passes %.3f"
          % (t.mean(), (t > 2).mean(), t[t > 2].sum() / t.sum()), flush=True)
r.close()
