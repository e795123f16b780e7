"""Diagnostic: ptr_render_device_noalloc recorded into a HIP graph and replayed; prints how the replayed frame differs."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, torch, ptr_amd
sc = json.load(open(os.path.join(ROOT, "tests", "golden", "scenes.json")))["example1"]
soa = ptr_amd.scene_to_soa(sc["elements"])
w, h = 320, 240
r = ptr_amd.Renderer(0)
r.set_scene(soa); r.set_background(sc["background"]); r.sync()
out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
ref = torch.zeros((h, w), dtype=torch.int32, device="cuda")
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
p = r.params(w, h, flags=flags)
side = torch.cuda.Stream()
# dirty the static array with a larger call first
big = torch.zeros((1080, 1920), dtype=torch.int32, device="cuda")
r.render_device_noalloc(r.params(1920, 1080, flags=flags), 0, 1, big.data_ptr(), stream=side.cuda_stream); side.synchronize()
r.render_device_noalloc(p, 0, 1, ref.data_ptr(), stream=side.cuda_stream); side.synchronize()
print("eager stats", r.stats())
r.render_device_noalloc(r.params(1920, 1080, flags=flags), 0, 1, big.data_ptr(), stream=side.cuda_stream); side.synchronize()
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.graph(g, stream=cap):
    r.render_device_noalloc(p, 0, 1, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
for k in range(3):
    out.fill_(0x5A5A5A5A)
    g.replay(); torch.cuda.synchronize()
    d = (out != ref)
    idx = d.reshape(-1).nonzero().reshape(-1)[:8].tolist()
    print("replay", k, "differ", int(d.sum()), idx, [hex(int(out.reshape(-1)[i]) & 0xffffffff) for i in idx], r.stats())
