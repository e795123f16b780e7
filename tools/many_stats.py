import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, torch, ptr_amd
from ptr_amd import synthetic
soa = synthetic.many_spheres()
r = ptr_amd.Renderer(0)
r.set_scene(soa); r.set_background([2.0, 2.0, 2.0])
out = torch.empty((1080, 1920), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, mode in (("hybrid", 0), ("tol", ptr_amd.MODE_TOLERANT)):
    p = r.params(1920, 1080, flags=mode | ptr_amd.FLAG_TIMING)
    for _ in range(3):
        r.render_device(p, 0, 1, out.data_ptr(), stream=s); torch.cuda.synchronize()
    print(name, r.kernel_ms(), r.stats())
