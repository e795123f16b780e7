"""Diagnostic: the pixels of one soak scene (tools/soak_ext_tol.py numbering) where the tolerant extension mode is beyond 1/255."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), os.path.join(ROOT, "tools")]
import numpy as np, ptr_amd
from soak_ext_tol import scene
target, seed, edge = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 2028, int(sys.argv[3]) if len(sys.argv) > 3 else 300
rng = np.random.default_rng(seed)
r = ptr_amd.Renderer(0)
for k in range(target + 1):
    soa, bg = scene(rng, k)
    w, h = int(rng.integers(9, edge)), int(rng.integers(5, edge))
    depth, spp, sd = int(rng.integers(0, 8)), int(rng.choice([1, 1, 1, 2, 3, 5])), int(rng.integers(0, 2**31))
    if depth == 0 and spp == 1: depth = 1
    cam = None
    if k % 4 == 3:
        org = rng.uniform(-6, 6, 3); fwd = rng.normal(size=3); fwd[2] = -abs(fwd[2]) - 0.5; fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
        cam = (org, right, np.cross(right, fwd), fwd)
if cam: r.set_camera(*cam)
else: r.set_camera(None)
r.set_scene(soa); r.set_background(bg)
np.set_printoptions(precision=4, suppress=True, linewidth=200)
print("scene", target, "spheres (cx cy cz r | sr sg sb | refl transp | er eg eb):"); print(soa.T); print("bg", bg, "size", w, h, "depth", depth, "spp", spp, "camera", cam)
for dd in range(depth + 1):
    want = r.render(w, h, flags=0, max_depth=dd, spp=spp, seed=sd)
    got = r.render(w, h, flags=ptr_amd.MODE_TOLERANT, max_depth=dd, spp=spp, seed=sd) if (dd or spp > 1) else want
    f32 = r.render(w, h, flags=ptr_amd.MODE_F32_ONLY, max_depth=dd, spp=spp, seed=sd) if (dd or spp > 1) else want
    d = np.abs(got.view(np.uint8).reshape(-1, 4).astype(np.int16) - want.view(np.uint8).reshape(-1, 4).astype(np.int16)).max(axis=1)
    idx = np.flatnonzero(d > 1)
    print(f"depth {dd}: tolerant beyond on {idx.size} px:", [(int(i % w), int(i // w), hex(int(want[i])), hex(int(got[i])), hex(int(f32[i]))) for i in idx[:8]])
