import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), ROOT]
import numpy as np, torch, ptr_amd
if os.environ.get("PTR_AB_LIB"):
    ptr_amd.LIB_PATH = os.path.abspath(os.environ["PTR_AB_LIB"])
from ptr_amd import synthetic as syn
from bench import example1_soa
r = ptr_amd.Renderer(0)
def t(label, flags, depth, frames=16, reps=5):
    out = torch.zeros((frames, 1080, 1920), dtype=torch.int32, device="cuda")
    p = r.params(1920, 1080, flags=flags, max_depth=depth)
    s = torch.cuda.current_stream().cuda_stream
    r.render_device(p, 0, frames, out.data_ptr(), stream=s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r.render_device(p, 0, frames, out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    print(f"{os.path.basename(ptr_amd.LIB_PATH)} {label}: {(time.perf_counter() - t0) / reps / frames * 1e3:.4f} ms per frame", flush=True)
soa, planes, phong = syn.planes_scene()
r.set_scene(np.ascontiguousarray(np.broadcast_to(soa, (16,) + soa.shape))); r.set_background([2.0, 2.0, 2.0]); r.set_extension(planes, phong)
for d in (0, 3, 5):
    t(f"planes+phong depth {d} f32", ptr_amd.MODE_F32_ONLY, d); t(f"planes+phong depth {d} f64", ptr_amd.MODE_EXACT_F64, d)
r.set_extension(None, None)
r.set_scene(np.ascontiguousarray(np.broadcast_to(example1_soa(), (16, 12, 7))))
for d in (3, 5):
    t(f"example1 (spheres only) depth {d} f32", ptr_amd.MODE_F32_ONLY, d); t(f"example1 depth {d} f64", ptr_amd.MODE_EXACT_F64, d)
r.close()
