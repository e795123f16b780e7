"""Where a host-buffer frame's time goes (GPU): ptr_render into a pageable buffer, into a pinned one, and the device-only call.
    python tools/host_path_time.py [width height reps]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), ROOT]
import numpy as np
import torch
import ptr_amd
import bench

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
soa = bench.example1_soa()
r = ptr_amd.Renderer(0)
r.set_scene(soa); r.set_background(np.array([2.0, 2.0, 2.0]))
p = r.params(w, h, flags=0)
L = r._L


def timed(f, n=reps):
    for _ in range(5): f()
    t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e3


pageable = np.empty(w * h, dtype=np.uint32)
pinned = torch.empty(w * h, dtype=torch.int32).pin_memory()
dev = torch.empty(w * h, dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()
pp = pageable.ctypes.data_as(C.POINTER(C.c_uint32))
pn = C.cast(pinned.data_ptr(), C.POINTER(C.c_uint32))
out = {
    "size": f"{w}x{h}",
    "ptr_render_pageable_ms": timed(lambda: r._check(L.ptr_render(r._h, C.byref(p), pp, None))),
    "ptr_render_pinned_ms": timed(lambda: r._check(L.ptr_render(r._h, C.byref(p), pn, None))),
    "fresh_numpy_each_call_ms": timed(lambda: r.render(w, h, flags=0)),
    "set_scene_plus_render_pinned_ms": timed(lambda: (r.set_scene(soa), r._check(L.ptr_render(r._h, C.byref(p), pn, None)))),
}


def device_only():
    r.render_device(p, 0, 1, dev.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()


out["render_device_sync_ms"] = timed(device_only)


def device_then_copy():
    r.render_device(p, 0, 1, dev.data_ptr(), stream=stream.cuda_stream)
    with torch.cuda.stream(stream):
        pinned.copy_(dev, non_blocking=True)
    stream.synchronize()


out["render_device_copy_pinned_ms"] = timed(device_then_copy)
t = time.perf_counter()
for _ in range(50):
    pageable[:] = pinned.numpy().view(np.uint32)
out["memcpy_pinned_to_pageable_ms"] = (time.perf_counter() - t) / 50 * 1e3
import json
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))
