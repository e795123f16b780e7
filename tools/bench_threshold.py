"""Which float32 kernel wins at which sphere count?  Times the SGPR-record kernel (ptr_kernels_fast.hip)
against the 4-pixel two-level kernel (ptr_kernels_many.hip, forced) on scenes of n spheres."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "py"))
import numpy as np, torch, ptr_amd
from ptr_amd import synthetic as syn
r = ptr_amd.Renderer(0)
W, H, F = 1920, 1080, 16
out = torch.empty(F * W * H, dtype=torch.int32, device="cuda")
for n in (7, 16, 32, 64, 96, 128):
    soa = syn.example1() if n == 7 else syn.many_spheres(n)
    r.set_scene(np.ascontiguousarray(np.broadcast_to(soa, (F,) + soa.shape))); r.set_background(syn.BACKGROUND)
    res = {}
    for name, fl in (("fast", 0), ("many", ptr_amd.DEBUG_FORCE_MANY)):
        p = r.params(W, H, flags=fl | ptr_amd.FLAG_TIMING)
        best = 1e9
        for _ in range(3):
            r.render_device(p, 0, F, out.data_ptr()); r.sync()
            k = r.kernel_ms(); best = min(best, k[0] + k[1])
        res[name] = round(best, 4)
    print(json.dumps(dict(n=n, frames=F, **res)), flush=True)
