import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, ptr_amd
if os.environ.get("PTR_AB_LIB"): ptr_amd.LIB_PATH = os.path.abspath(os.environ["PTR_AB_LIB"])
from bench import example1_soa
r = ptr_amd.Renderer(0); r.set_scene(example1_soa()); r.set_background([2.0, 2.0, 2.0])
for (w, h, d) in ((1920, 1080, 5), (3840, 2160, 5), (1920, 1080, 3)):
    a = r.render(w, h, flags=ptr_amd.MODE_F32_ONLY, max_depth=d); b = r.render(w, h, flags=0, max_depth=d)
    dd = np.abs(a.view(np.uint8).astype(np.int16) - b.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
    print(os.path.basename(ptr_amd.LIB_PATH), w, h, d, "beyond 1/255:", int((dd > 1).sum()), "worst", int(dd.max()))
