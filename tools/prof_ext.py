"""Small fixed workload for rocprofv3: the float32 extension kernel, example scene, 1080p, depth 5.
Usage: python tools/prof_ext.py [steps] [frames] [depth] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, torch, ptr_amd
from ptr_amd import synthetic as syn
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 5
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
soa = syn.example1()
r = ptr_amd.Renderer(0)
r.set_scene(np.ascontiguousarray(np.broadcast_to(soa, (frames,) + soa.shape))); r.set_background(syn.BACKGROUND)
out = torch.empty((frames, 1080, 1920), dtype=torch.int32, device="cuda")
p = r.params(1920, 1080, flags=ptr_amd.MODE_F32_ONLY | ptr_amd.FLAG_TIMING, max_depth=depth, spp=spp, seed=1)
for _ in range(steps):
    r.render_device(p, 0, frames, out.data_ptr()); r.sync()
    print("kernel ms:", r.kernel_ms()[0])
