"""Re-trace kernel diagnostics (GPU): queue fill of the example scene at 1080p and the kernels' time.
Usage: python tools/diag_retrace.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, torch, ptr_amd
from bench import example1_soa, WIDTH, HEIGHT
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
soa = example1_soa()
r = ptr_amd.Renderer(0)
r.set_scene(np.ascontiguousarray(np.broadcast_to(soa, (frames,) + soa.shape))); r.set_background([2.0, 2.0, 2.0])
out = torch.empty((frames, HEIGHT, WIDTH), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
r.render_device(r.params(WIDTH, HEIGHT, flags=0), 0, 1, out.data_ptr(), stream=s); torch.cuda.synchronize()
st = r.stats(); print("one frame: retraced %d  reshaded %d of %d pixels" % (st["retraced"], st["reshaded"], WIDTH * HEIGHT))
p = r.params(WIDTH, HEIGHT, flags=ptr_amd.FLAG_TIMING)
ms = []
for _ in range(6):
    r.render_device(p, 0, frames, out.data_ptr(), stream=s); torch.cuda.synchronize(); ms.append(r.kernel_ms())
print("f32 %.4f ms  re-trace %.4f ms (min %.4f)" % (np.median([m[0] for m in ms[1:]]), np.median([m[1] for m in ms[1:]]), min(m[1] for m in ms[1:])))
