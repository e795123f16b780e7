"""One extension-mode call per mode name (f32 | exact), example scene, 1080p, depth 5: something short to put under rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), ROOT]
import numpy as np, torch, ptr_amd
from bench import example1_soa
mode = {"f32": ptr_amd.MODE_F32_ONLY, "exact": ptr_amd.MODE_EXACT_F64}[sys.argv[1]]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 5
r = ptr_amd.Renderer(0)
r.set_scene(example1_soa()); r.set_background([2.0, 2.0, 2.0])
out = torch.zeros((1, 1080, 1920), dtype=torch.int32, device="cuda")
p = r.params(1920, 1080, flags=mode, max_depth=depth)
for _ in range(3):
    r.render_device(p, 0, 1, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
r.close()
