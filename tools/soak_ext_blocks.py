"""Soak test (GPU): the float32 bounce loop with more than 32 samples per pixel - a wave per block of samples + the ordered sum
(render_ext_f32_split_kernel + ext_combine_kernel) against the same sums kept in registers by one wave (the entry point that
allocates nothing: render_ext_f32_blocks_kernel), on random scenes, sizes, bands, depths and sample counts: bytes AND float32
colours must be identical; beside it the PSNR against the float64 specification.   python tools/soak_ext_blocks.py [scenes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, torch
import ptr_amd

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3232)
F, X = ptr_amd.MODE_F32_ONLY, ptr_amd.MODE_EXACT_F64
r = ptr_amd.Renderer(0)
s = torch.cuda.current_stream().cuda_stream
bad = px = 0
worst = 1e9
t0 = time.time()
for k in range(n_scenes):
    n = int(rng.integers(1, 13)) if k % 6 else int(rng.integers(13, 40))
    soa = np.zeros((12, n))
    soa[0] = rng.uniform(-9, 9, n); soa[1] = rng.uniform(-3, 6, n); soa[2] = rng.uniform(-45, -6, n)
    soa[3] = rng.uniform(0.2, 4.5, n); soa[4:7] = rng.uniform(0.0, 1.0, (3, n))
    soa[7] = (rng.random(n) < 0.6) * rng.uniform(0.1, 1.0, n); soa[8] = (rng.random(n) < 0.4) * rng.uniform(0.1, 1.0, n)
    lights = rng.random(n) < 0.25; lights[rng.integers(0, n)] = True
    soa[9:12, lights] = rng.uniform(0.1, 2.5, (3, int(lights.sum())))
    if k % 3 == 0:
        soa[:, 0] = [0, -10000 - rng.uniform(2, 6), -20, 10000, 0.2, 0.2, 0.2, float(k % 6 == 0), 0, 0, 0, 0]
    planes = None
    if k % 4 == 1:
        planes = np.zeros((14, 1)); planes[0:3, 0] = [0.0, -4.0 - rng.uniform(0, 3), -20.0]; planes[3:6, 0] = [0, 1, 0]
        planes[6:9, 0] = rng.uniform(0.1, 0.9, 3)
    r.set_scene(np.ascontiguousarray(soa)); r.set_background(rng.uniform(0, 2.0, 3)); r.set_extension(planes, None)
    w, h = int(rng.integers(9, 160)), int(rng.integers(5, 110))
    y0 = int(rng.integers(0, h // 2)) if k % 5 == 0 else 0
    rows = int(rng.integers(1, h - y0 + 1)) if k % 5 == 0 else h
    depth, spp, seed = int(rng.integers(0, 6)), int(rng.choice([33, 40, 64, 65, 96, 100, 129, 200])), int(rng.integers(0, 2**31))
    p = r.params(w, h, y0, rows, flags=F, max_depth=depth, spp=spp, seed=seed)
    a8 = torch.zeros((rows, w), dtype=torch.int32, device="cuda"); af = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
    b8 = torch.zeros_like(a8); bf = torch.zeros_like(af)
    r.render_device(p, 0, 1, a8.data_ptr(), af.data_ptr(), stream=s); r.sync()
    r.render_device_noalloc(p, 0, 1, b8.data_ptr(), bf.data_ptr(), stream=s); torch.cuda.synchronize()
    d = int((a8 != b8).sum().item()) + int((af != bf).any(dim=-1).sum().item())
    bad += d; px += rows * w
    if k % 10 == 0:
        want = r.render(w, h, y0, rows, flags=X, max_depth=depth, spp=spp, seed=seed).view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
        got = a8.cpu().numpy().view(np.uint8).reshape(-1, 4)[:, :3].astype(np.float64)
        mse = float(((want - got) ** 2).mean())
        worst = min(worst, 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse))
    if d or k % 100 == 0:
        print(f"scene {k}: {n} spheres {w}x{rows} of {h} depth {depth} spp {spp}: {d} differ   [{time.time() - t0:.0f} s]", flush=True)
print(f"{n_scenes} scenes, {px} pixels: the two forms of the block sums differ on {bad} pixels; worst PSNR of the float32 form against float64 {worst:.1f} dB")
sys.exit(1 if bad else 0)
