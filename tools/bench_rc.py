"""Measures the render cache's reprojection pass (ptr_rc_reproject_device) on one MI355X: 1080p cache, 4 items
per pixel (Driver.cacheFactor), inputs resident in HBM.  Prints one JSON line with the HBM roofline:
algorithmic bytes = per item 24 B hit + 8 B age read + 8 B age written + 4 B item_pixel written,
per interior pixel 4 B element + 8 B depth written (two writes each: reset + result are not counted twice)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "py"))
import numpy as np, torch, ptr_amd

W, H, FACTOR = 1920, 1080, 4
n = W * H * FACTOR
rng = np.random.default_rng(7)
hits = np.stack([rng.uniform(-30, 30, n), rng.uniform(-6, 14, n), rng.uniform(-60, -8, n)], axis=1)
hits[rng.random(n) < 0.2, 0] = np.nan
# camera of tests/golden/rc/reproject_move_96x64 with a 16:9 projection
rec = np.fromfile(os.path.join(ROOT, "tests", "golden", "rc", "reproject_move_96x64.f64"), count=35)
proj = rec[19:35].copy(); proj[0] = proj[5] / (W / H)
r = ptr_amd.Renderer(0)
cam = r.rc_camera(rec[0:3], rec[3:19], proj, W, H)
d_hit = torch.from_numpy(hits).cuda()
d_age0 = torch.from_numpy(rng.integers(0, 30, n).astype(np.float64)).cuda()
d_age = d_age0.clone()
d_item = torch.empty(n, dtype=torch.int32, device="cuda")
cells = (W + 2) * (H + 2)
d_pel = torch.full((cells,), -2, dtype=torch.int32, device="cuda")
d_pdep = torch.zeros(cells, dtype=torch.float64, device="cuda")
s = None          # the context's own stream; timed with the wall clock around ptr_sync()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    r.rc_reproject_device(cam, n, d_hit.data_ptr(), d_age.data_ptr(), d_item.data_ptr(), d_pel.data_ptr(), d_pdep.data_ptr(), stream=s)
r.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    r.rc_reproject_device(cam, n, d_hit.data_ptr(), d_age.data_ptr(), d_item.data_ptr(), d_pel.data_ptr(), d_pdep.data_ptr(), stream=s)
r.sync()
ms = (time.perf_counter() - t0) * 1e3 / steps
# ---- rank 3 on the state the reprojection left: depth culling + gap filling + the three frames
import ctypes as C
d_rgb = torch.from_numpy(rng.integers(0, 1 << 24, n).astype(np.uint32).view(np.int32)).cuda()
d_prgb = torch.zeros(cells, dtype=torch.int32, device="cuda"); d_pprio = torch.zeros(cells, dtype=torch.int32, device="cuda")
d_hist = torch.zeros(768, dtype=torch.int32, device="cuda"); d_tot = torch.zeros(3, dtype=torch.int64, device="cuda")
d_frames = torch.zeros(3 * W * H, dtype=torch.int32, device="cuda")
sp = ptr_amd.RcShadeParams(W, H, 1.0, 20.0, 5.0, 255.0)
def shade():
    r._check(r._L.ptr_rc_shade_device(r._h, C.byref(sp), n, d_pel.data_ptr(), d_pdep.data_ptr(), d_age.data_ptr(), d_rgb.data_ptr(),
                                      d_prgb.data_ptr(), d_pprio.data_ptr(), d_hist.data_ptr(), d_tot.data_ptr(),
                                      d_frames.data_ptr(), d_frames.data_ptr() + 4 * W * H, d_frames.data_ptr() + 8 * W * H, None))
for _ in range(3): shade()
r.sync()
t0 = time.perf_counter()
for _ in range(steps): shade()
r.sync()
ms2 = (time.perf_counter() - t0) * 1e3 / steps
alg2 = W * H * (4 + 8 + 8 + 4 + 4 + 4 + 12)      # element, depth, the element's age and colour, colour + priority out, 3 frames
print(json.dumps({"metric": "render-cache depth culling + gap filling + frames, pixels/s", "value": round(W * H / ms2 / 1e3, 1), "unit": "Mpixels/s",
                  "ms_per_pass": round(ms2, 4), "frame": f"{W}x{H}", "pixels_with_element": int(d_tot[0].item()),
                  "roofline": {"bound": "hbm", "achieved": round(alg2 / ms2 / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(alg2 / ms2 / 1e6 / 8000.0, 4), "algorithmic_bytes": alg2}}))
alg = n * (24 + 8 + 8 + 4) + W * H * 12
print(json.dumps({"metric": "render-cache reprojection, cache items/s", "value": round(n / ms / 1e3, 1), "unit": "Mitems/s",
                  "ms_per_pass": round(ms, 4), "items": n, "frame": f"{W}x{H}", "attached": int((d_item >= 0).sum().item()),
                  "dtype": "f64", "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                                                "frac": round(alg / ms / 1e6 / 8000.0, 4), "algorithmic_bytes": alg}}))
