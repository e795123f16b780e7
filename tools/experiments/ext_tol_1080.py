"""BASELINE configs[1] (example scene, 1080p, depth 5) in the three precision forms of the extension modes: bytes against the
float64 specification, share re-done, time per 64-frame batch.   python tools/ext_tol_1080.py [frames] [depth]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, torch, ptr_amd
from bench import example1_soa, WIDTH, HEIGHT
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 5
soa = example1_soa()
r = ptr_amd.Renderer(0)
r.set_scene(np.ascontiguousarray(np.broadcast_to(soa, (frames,) + soa.shape))); r.set_background([2.0, 2.0, 2.0])
out = torch.empty((frames, HEIGHT, WIDTH), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
res = {}
for name, fl in (("float64", 0), ("tolerant", ptr_amd.MODE_TOLERANT), ("float32", ptr_amd.MODE_F32_ONLY)):
    p = r.params(WIDTH, HEIGHT, flags=fl, max_depth=depth)
    for _ in range(2): r.render_device(p, 0, frames, out.data_ptr(), stream=s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3 if name == "float64" else 6
    for _ in range(n): r.render_device(p, 0, frames, out.data_ptr(), stream=s)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    res[name] = out[0].cpu().numpy().view(np.uint32).ravel().copy()
    same = bool(torch.equal(out[0], out[frames - 1]))
    st = r.stats()
    d = np.abs(res[name].view(np.uint8).reshape(-1, 4).astype(np.int16) - res["float64"].view(np.uint8).reshape(-1, 4).astype(np.int16)).max(axis=1)
    print(f"{name:9s} {ms:8.3f} ms per {frames} frames ({ms / frames:.4f} ms per frame); beyond 1/255: {int((d > 1).sum())} px, differ: {int((d > 0).sum())}; "
          f"re-done {st['retraced']} px of the call ({100.0 * st['retraced'] / (frames * WIDTH * HEIGHT):.2f} %); frames equal: {same}")
