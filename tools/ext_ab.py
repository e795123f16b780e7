"""Times the float32 extension kernel (PTR_MODE_F32_ONLY; PTR_AB_MODE=exact / lanewise: the float64 ones) on one MI355X and prints a hash of what it drew, so that two builds
of the library can be compared on one box:  python tools/ext_ab.py  (the library in the tree);
PTR_AB_LIB=path/to/other.so python tools/ext_ab.py  (another build)."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), ROOT]
import numpy as np, torch, ptr_amd
if os.environ.get("PTR_AB_LIB"):
    ptr_amd.LIB_PATH = os.path.abspath(os.environ["PTR_AB_LIB"])
from ptr_amd import synthetic as syn
from bench import example1_soa

MODE = {"exact": ptr_amd.MODE_EXACT_F64, "lanewise": ptr_amd.MODE_EXACT_F64 | ptr_amd.DEBUG_EXT_LANEWISE}.get(os.environ.get("PTR_AB_MODE"), ptr_amd.MODE_F32_ONLY)
# (PTR_AB_MODE=lanewise: the float64 form by the lane-by-lane kernel, PTR_DEBUG_EXT_LANEWISE; exact: the wave-level one)

def run(r, name, w, h, depth, spp, frames, reps):
    out = torch.zeros((frames, h, w), dtype=torch.int32, device="cuda")
    p = r.params(w, h, flags=MODE, max_depth=depth, spp=spp, seed=0x5EED1234)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        r.render_device(p, 0, frames, out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r.render_device(p, 0, frames, out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"case": name, "ms": round(ms, 4), "sha": sha, "lib": os.path.basename(ptr_amd.LIB_PATH)}), flush=True)

r = ptr_amd.Renderer(0)
r.set_scene(np.ascontiguousarray(np.broadcast_to(example1_soa(), (256, 12, 7)))); r.set_background([2.0, 2.0, 2.0])
run(r, "example1 1080p depth5 x256", 1920, 1080, 5, 1, 256 if MODE == ptr_amd.MODE_F32_ONLY else 64, 5)
run(r, "example1 1080p depth3 x64", 1920, 1080, 3, 1, 64, 5)
run(r, "example1 640x480 depth5 x1", 640, 480, 5, 1, 1, 20)
r.set_scene(syn.config3_scene()); r.set_background(syn.BACKGROUND)
run(r, "config2 1080p depth5 64spp", 1920, 1080, 5, 64, 1, 3)
run(r, "config3 4K depth5 16spp", 3840, 2160, 5, 16, 1, 3)
r.close()
