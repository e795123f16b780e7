"""BASELINE configs[3] on one GPU: the 8 parts of an 8-GPU split of a 4K / 256 spp / depth 5 frame one after the other (what
bench.py's config3_4k_256spp times), and configs[2] (1080p, 64 spp); float32 form.  PTR_AB_LIB=other.so for another build."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), ROOT]
import numpy as np, torch, ptr_amd
if os.environ.get("PTR_AB_LIB"):
    ptr_amd.LIB_PATH = os.path.abspath(os.environ["PTR_AB_LIB"])
from ptr_amd import synthetic as syn
r = ptr_amd.Renderer(0)
r.set_scene(syn.config3_scene()); r.set_background(syn.BACKGROUND)
s = torch.cuda.current_stream().cuda_stream
W, H, F = 3840, 2160, ptr_amd.MODE_F32_ONLY
parts = [ptr_amd.StripPart(8, k, 8) for k in range(8)]
buf = torch.empty((parts[0].local_rows(H), W), dtype=torch.int32, device="cuda")


def frame():
    for sp in parts:
        rows = sp.local_rows(H)
        r.render_strips_device(r.params(W, H, 0, rows, flags=F, max_depth=5, spp=256, seed=0x5EED1234), sp, 0, 1, buf.data_ptr(), stream=s)
    torch.cuda.synchronize()


def timed(f, n):
    f(); f()
    t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e3


out = {"lib": os.path.basename(ptr_amd.LIB_PATH), "config3_ms_per_frame": round(timed(frame, 5), 3)}
b2 = torch.empty((1080, 1920), dtype=torch.int32, device="cuda")
p2 = r.params(1920, 1080, flags=F, max_depth=5, spp=64, seed=0x5EED1234)


def c2():
    r.render_device(p2, 0, 1, b2.data_ptr(), stream=s); torch.cuda.synchronize()


out["config2_ms_per_frame"] = round(timed(c2, 10), 4)
print(json.dumps(out))
