"""Small fixed workload for rocprofv3: N launches of the bench step (64 frames of the example
scene at 1080p, hybrid mode).  Usage: python tools/prof_run.py [steps] [frames] [mode] [example1|many]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np
import torch
import ptr_amd
from bench import example1_soa, WIDTH, HEIGHT

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mode = {"hybrid": 0, "tol": ptr_amd.MODE_TOLERANT, "f32": ptr_amd.MODE_F32_ONLY, "exact": ptr_amd.MODE_EXACT_F64,
        "hybrid_many": ptr_amd.DEBUG_FORCE_MANY, "f32_many": ptr_amd.MODE_F32_ONLY | ptr_amd.DEBUG_FORCE_MANY}[sys.argv[3] if len(sys.argv) > 3 else "hybrid"]
soa = example1_soa()
if len(sys.argv) > 4 and sys.argv[4] == "many":
    from ptr_amd import synthetic
    soa = synthetic.many_spheres()
r = ptr_amd.Renderer(0)
r.set_scene(np.ascontiguousarray(np.broadcast_to(soa, (frames,) + soa.shape)))
r.set_background([2.0, 2.0, 2.0])
out = torch.empty((frames, HEIGHT, WIDTH), dtype=torch.int32, device="cuda")
depth = int(os.environ.get("PTR_PROF_DEPTH", "0"))       # > 0: the extension kernel at that depth
p = r.params(WIDTH, HEIGHT, flags=mode | ptr_amd.FLAG_TIMING, max_depth=depth)
s = torch.cuda.current_stream().cuda_stream
ms = []
for _ in range(steps):
    r.render_device(p, 0, frames, out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    ms.append(r.kernel_ms())
# whole steps back to back (no events, no sync in between): what bench.py times
import time
p2 = r.params(WIDTH, HEIGHT, flags=mode, max_depth=depth)
for _ in range(3):
    r.render_device(p2, 0, frames, out.data_ptr(), stream=s)
torch.cuda.synchronize()
n2 = max(steps, 20)
t0 = time.perf_counter()
for _ in range(n2):
    r.render_device(p2, 0, frames, out.data_ptr(), stream=s)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n2 * 1e3
print("kernel ms (main, retrace):", ms, " wall ms per step: %.4f" % wall, " Mpixel/s:", frames * WIDTH * HEIGHT / (ms[-1][0] * 1e-3) / 1e6)
