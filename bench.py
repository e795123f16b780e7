#!/usr/bin/env python3
"""bench.py — throughput of the per-pixel hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (camera ray -> nearest hit -> shadowed Lambert -> RGBA8 pack,
RT/src/RayTracer.js:163-193) over one batch: FRAMES frames of the reference's example scene
(RT/examples/example1.js:9-45) at 1920x1080, 1 sample per pixel, reference semantics (the reference
has no bounce loop).  Precision mode of the headline: PTR_MODE_TOLERANT (include/ptr.h) - float32 with the
discontinuous doubts re-traced in float64, every channel of every pixel within 1/255 of the reference (north_star's
bar; tests/test_gpu_tolerant.py); the default mode of the library, whose output BYTES equal the reference's, is timed
beside it on the same workload ("bit_exact" in the line; --mode exact makes it the headline).
Scene blocks are resident in HBM before the timed region and the frames stay in HBM.

N > 1: every frame is split into cyclic strips of 8 scanlines over the ranks (the reference's
RenderPlanner band split, RT/src/RenderPlanner.js:75-76, made cyclic for load balance); every
step's strips are gathered into whole frames on rank 0 over RCCL by the library's own gather
(ptr_gather_strips_device: grouped ncclSend / ncclRecv straight over each peer's xGMI link, then one
copy kernel on rank 0), on a communication stream, double-buffered, so that step k's gather runs
under step k + 1's render; all of it inside the timed region (the last gather included).  The first
gathered frame is checked against a single-GPU render before anything is timed; if the library's
gather cannot be used (ranks sharing a GPU in a rehearsal, no RCCL) the bench says so ("gather":
"torch") and gathers with torch.distributed + the library's copy kernel instead.  Weak scaling:
each rank renders FRAMES * N frames' worth of its strips, i.e. per-GPU work is fixed as N grows.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "py"))

WIDTH, HEIGHT = 1920, 1080
FRAMES_PER_GPU = 256         # per launch: 2.1 GB of RGBA8 per GPU; 64 -> 256 frames per launch is +11 % (launch ramp and gaps amortised)
STRIP = 8
PEAK_FP32_VALU_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
PEAK_HBM_GBPS = 8000.0
# HBM traffic of the float32 kernel per pixel, from the PMC passes of this very command kept under profiles/
# (r03_summary.md, kernel render_fast_kernel<2, false, 4, false, false>: WRITE_SIZE 2 092 250 KB and FETCH_SIZE 4 110.0 KB per
# launch of 256 x 1920 x 1080 pixels; render_fast_kernel<1, false, 3, ...> of --mode exact: 2 306 760 KB and 5 768.0 KB; gfx950
# correction of the guide applied: FETCH_SIZE counts 64 B per 128-B request -> x 2).  Counters cannot be read from
# inside this process, so the per-launch figure reported below is this profiled rate times the launch's pixels
# (reported as "traffic", with "traffic_source" saying so).
PROFILED_HBM_BYTES_PER_PIXEL = {"tolerant": (2092070.0 * 1024 + 2 * 4113.06 * 1024) / (256 * 1920 * 1080),
                                "exact": (2306920.0 * 1024 + 2 * 5707.05 * 1024) / (256 * 1920 * 1080)}
# ... and of one reprojection pass over the 1920 x 1080 x 4 cache on one GPU (same profile: rc_clear + rc_project + hipCUB's five
# radix-sort kernels + rc_resolve; FETCH_SIZE 11.5 + 129 619 + 5 x 33 011.6 + 328 372 KB, x 2; WRITE_SIZE 24 840 + 209 808 +
# 5 x 60 016.5 + 142 818 KB): 1.97 GB against 0.39 GB algorithmic
PROFILED_RC_BYTES_PER_PASS = int((2 * (11.5 + 129619 + 5 * 33011.6 + 328372) + (24840 + 209808 + 5 * 60016.5 + 142818)) * 1024)


def example1_soa():
    """RT/examples/example1.js:9-45 as the 12-plane SoA of include/ptr.h."""
    rows = [  # cx, cy, cz, radius, sr, sg, sb, refl, transp, er, eg, eb
        (0.0, -10004, -20, 10000, 0.2, 0.2, 0.2, 0, 0, 0, 0, 0),
        (0, 0, -20, 4, 1.00, 0.32, 0.36, 1, 0.5, 0, 0, 0),
        (5, -1, -15, 2, 0.9, 0.76, 0.46, 1, 0, 0, 0, 0),
        (5, 0, -25, 3, 0.65, 0.77, 0.97, 1, 0, 0, 0, 0),
        (-5.5, 0, -15, 3, 0.9, 0.9, 0.9, 1, 0, 0, 0, 0),
        (0, 20, -30, 3, 0, 0, 0, 0, 0, 1.2, 1.2, 1.2),
        (0, 10, 10, 3, 0, 0, 0, 0, 0, 1, 1, 1),
    ]
    return np.ascontiguousarray(np.array(rows, dtype=np.float64).T)


def algorithmic_flops(st):
    """30 P + 8 A + 15 B + 19 C + 33 H + 31 (hit, light) pairs."""
    return (30 * st["primary"] + 8 * st["exit_tca"] + 15 * st["exit_d2"] + 19 * st["exit_hit"]
            + 33 * st["hits"] + 31 * st["shadow_rays"])


def algorithmic_flops_ext(st, n_lights):
    """The same counting rule for the extension modes' specification (include/ptr.h), from the work counters of the
    float64 counting run: 30 per pixel (ray generation + pack); every traced ray's sphere tests 8 / 15 / 19 by exit; per
    hit 24 (hit frame) + 6 (w * emission); per miss 6 (w * background); a hit that shades: 6 (shadow origin) + 31 per
    light + 6 (w * sum); a hit that bounces: 11 (facing, fresnel, w * surfaceColor) + 38 for the reflection ray
    (direction 7, normalize 10, origin 6, weight 3, ...) and 43 for a refraction ray.  Derived counts: shading hits =
    shadow rays / lights, bouncing hits = hits - shading hits = reflection rays, refraction rays = secondary - reflection."""
    rays = st["primary"] + st["secondary"]
    hits = st["hits"]
    shading = st["shadow_rays"] // max(1, n_lights)
    bouncing = hits - shading
    refraction = st["secondary"] - bouncing
    return (30 * st["primary"] + 8 * st["exit_tca"] + 15 * st["exit_d2"] + 19 * st["exit_hit"] + 30 * hits + 6 * (rays - hits)
            + 12 * shading + 31 * st["shadow_rays"] + 49 * bouncing + 43 * refraction)


def cpu_baseline(soa, bg, seconds=12.0):
    """The CPU oracle (float64 C restatement of the reference) on this host's cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as orc                      # checker / baseline only, never the product path
    threads = max(1, min(16, os.cpu_count() or 1))
    orc.render(soa, bg, 320, 180, flags=0, threads=threads)          # warm up, page in
    _, cnt = orc.render(soa, bg, WIDTH, HEIGHT, flags=0, counters=True)
    rays = cnt["primary"] + cnt["shadow_rays"]
    t0 = time.perf_counter()
    frames = 0
    while True:
        orc.render(soa, bg, WIDTH, HEIGHT, flags=0, threads=threads)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": round(rays * frames / dt / 1e6, 2), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} frames of the example scene at {WIDTH}x{HEIGHT} in {dt:.1f} s, "
                      f"oracle/ptr_oracle.c (float64 C port of the reference) on {threads} pthreads"}


def cpu_baseline_js(seconds=8.0):
    """The JavaScript CPU port (oracle/js, the reference's style and speed class) under Node worker_threads,
    one Worker per scanline band as the reference's RenderPlanner does (RT/src/RenderPlanner.js:75-76)."""
    import shutil
    import subprocess
    node = shutil.which("node")
    if not node:
        return {"value": None, "note": "no node on this host"}
    threads = max(1, min(16, os.cpu_count() or 1))
    try:
        res = subprocess.run([node, os.path.join(ROOT, "oracle", "js", "cpu_bench.mjs"), str(threads), str(seconds)],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=seconds * 6 + 60, check=True)
        r = json.loads(res.stdout.decode().strip().splitlines()[-1])
    except Exception as e:                       # a baseline, not the product: report and go on
        return {"value": None, "note": f"node run failed: {e}"}
    return {"value": round(r["mrays_per_s"], 2), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{r['frames']} frames of the example scene at {WIDTH}x{HEIGHT} in {r['seconds']:.1f} s, "
                      f"oracle/js/cpu_port.mjs (JavaScript port in the reference's object style) on {threads} "
                      f"worker_threads, node {r['node']}",
            "unmodified_reference_in_build_container": {
                "value": 35.6, "unit": "Mrays/s", "cores": 8, "one_thread": 5.29,
                "note": "not measured in this run: the UNMODIFIED reference under Node 12 worker_threads in the build container "
                        "(BASELINE.md); it cannot travel to the GPU box, which is why the figures measured here are ports"}}


def rc_reprojection_line(ptr_amd, torch, dist, rdr, world, rank, backend, library_comm, stream, steps):
    """BASELINE.json configs[4], second half: the render cache's point reprojection (rendercache-js Driver.reprojectFrame) of a
    1080p cache (4 items per pixel, synthetic hit points), the cache's items split over the ranks in cache order (strong scaling:
    the cache is the same at every N).  Every rank: local pass, all-gather of the local grids (12 B per pixel and rank), merge.
    Checked before it is timed: rank 0 also runs the whole cache alone and compares the grid and its own items."""
    W, H, FACTOR = 1920, 1080, 4
    n, cells = W * H * FACTOR, (W + 2) * (H + 2)
    rng = np.random.default_rng(7)
    hits = np.stack([rng.uniform(-30, 30, n), rng.uniform(-6, 14, n), rng.uniform(-60, -8, n)], axis=1)
    hits[rng.random(n) < 0.2, 0] = np.nan
    age0 = rng.integers(0, 30, n).astype(np.float64)
    rec = np.fromfile(os.path.join(ROOT, "tests", "golden", "rc", "reproject_move_96x64.f64"), count=35)
    proj = rec[19:35].copy(); proj[0] = proj[5] / (W / H)
    cam = rdr.rc_camera(rec[0:3], rec[3:19], proj, W, H)
    lo, hi = n * rank // world, n * (rank + 1) // world
    m = hi - lo
    d_hit = torch.from_numpy(hits[lo:hi]).cuda()
    d_age0 = torch.from_numpy(age0[lo:hi]).cuda()
    d_age = d_age0.clone()
    d_ip = torch.empty(m, dtype=torch.int32, device="cuda")
    d_pel = torch.full((cells,), -1, dtype=torch.int32, device="cuda")
    d_pdep = torch.full((cells,), 1e6, dtype=torch.float64, device="cuda")
    all_el = torch.full((world, cells), -1, dtype=torch.int32, device="cuda")
    all_dep = torch.full((world, cells), 1e6, dtype=torch.float64, device="cuda")
    mode = ["one GPU" if world == 1 else ("library (ncclAllGather)" if library_comm else "torch.distributed all_gather")]
    dev = "cuda" if backend == "nccl" else "cpu"

    def one_pass():
        if world == 1:
            rdr.rc_reproject_device(cam, n, d_hit.data_ptr(), d_age.data_ptr(), d_ip.data_ptr(), d_pel.data_ptr(), d_pdep.data_ptr(), stream=stream)
        elif mode[0].startswith("library"):
            rdr.rc_reproject_multi_device(cam, m, lo, d_hit.data_ptr(), d_age.data_ptr(), d_ip.data_ptr(), d_pel.data_ptr(), d_pdep.data_ptr(),
                                          stream=stream)
        else:
            rdr.rc_reproject_shard_device(cam, m, lo, d_hit.data_ptr(), d_age.data_ptr(), d_ip.data_ptr(), all_el[rank].data_ptr(),
                                          all_dep[rank].data_ptr(), stream=stream)
            if backend == "nccl":
                dist.all_gather_into_tensor(all_el, all_el[rank].clone())
                dist.all_gather_into_tensor(all_dep, all_dep[rank].clone())
            else:                                      # rehearsal: through the host
                torch.cuda.synchronize()
                for t in (all_el, all_dep):
                    parts = [torch.empty(cells, dtype=t.dtype) for _ in range(world)]
                    dist.all_gather(parts, t[rank].cpu())
                    t.copy_(torch.stack(parts))
            rdr.rc_merge_shards_device(W, H, world, rank, all_el.data_ptr(), all_dep.data_ptr(), m, lo, d_age.data_ptr(), d_ip.data_ptr(),
                                       d_pel.data_ptr(), d_pdep.data_ptr(), stream=stream)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def checked_pass():
        """-> (raised, good) agreed by all ranks"""
        d_age.copy_(d_age0)
        err = 0
        try:
            one_pass()
            torch.cuda.synchronize()
        except ptr_amd.PtrError as e:
            print(f"rank {rank}: reprojection: {e}", file=sys.stderr)
            err = 1
        if world > 1:
            flag = torch.tensor([err], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            err = int(flag.item())
        good = 1
        if rank == 0 and not err:
            r1 = ptr_amd.Renderer(torch.cuda.current_device())
            w_hit, w_age = torch.from_numpy(hits).cuda(), torch.from_numpy(age0).cuda()
            w_ip = torch.empty(n, dtype=torch.int32, device="cuda")
            w_pel = torch.full((cells,), -1, dtype=torch.int32, device="cuda")
            w_pdep = torch.full((cells,), 1e6, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            r1.rc_reproject_device(cam, n, w_hit.data_ptr(), w_age.data_ptr(), w_ip.data_ptr(), w_pel.data_ptr(), w_pdep.data_ptr())
            r1.sync()
            inner = torch.zeros((H + 2, W + 2), dtype=torch.bool, device="cuda"); inner[1:-1, 1:-1] = True
            inner = inner.reshape(-1)
            good = int(torch.equal(w_pel[inner], d_pel[inner]) and torch.equal(w_pdep[inner], d_pdep[inner]) and
                       torch.equal(w_age[lo:hi], d_age) and torch.equal(w_ip[lo:hi], d_ip))
            r1.close()
        if world > 1:
            flag = torch.tensor([good], dtype=torch.int32, device=dev)
            dist.broadcast(flag, src=0)
            good = int(flag.item())
        return err, good

    err, good = checked_pass()
    if (err or not good) and mode[0].startswith("library"):
        mode[0] = "torch.distributed all_gather (the library's exchange " + ("raised an error)" if err else "failed its check)")
        err, good = checked_pass()
    if err or not good:
        return {"error": "the sharded pass does not reproduce the single-GPU pass; not timed"}
    for _ in range(2):
        one_pass()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / steps * 1e3
    alg = n * (24 + 8 + 8 + 4) + W * H * 12          # per item: hit, age read + written, pixel pointer; per pixel: element + depth
    return {"workload": f"BASELINE.json configs[4]: render-cache point reprojection, {W}x{H} cache of {n} items"
                        + (f" split over {world} GPUs in cache order" if world > 1 else ""),
            "value": round(n / ms / 1e3, 1), "unit": "Mitems/s", "ms_per_pass": round(ms, 4), "n_gpus": world, "scaling": "strong",
            "dtype": "f64", "exchange": mode[0], "exchanged_bytes_into_each_rank": int(12 * cells * (world - 1)),
            "checked": "grid and rank 0's items equal the single-GPU pass over the whole cache",
            "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(alg / ms / 1e6 / PEAK_HBM_GBPS, 4), "algorithmic_bytes": alg,
                         "traffic": PROFILED_RC_BYTES_PER_PASS if world == 1 else None,
                         "traffic_source": "not measured in this run: PMC bytes of one pass in the profiled run of this command "
                                           "(profiles/r03_summary.md, WRITE_SIZE + 2 x FETCH_SIZE over the pass's nine kernels); "
                                           "5 x the algorithmic bytes: the three radix-sort passes and the gathers by cache index "
                                           "(profiles/r03_rc_reproject.md)" if world == 1 else "no profile of the sharded pass",
                         "note": "algorithmic bytes of the whole cache / the wall time of a pass (all ranks, exchange included)"}}


def exchange_probe_child(uid_hex, rank, world, device):
    """Child process of one rank (N > 1): the library's communicator and one small checked gather, away from the bench's own
    process - a collective that never returns then hangs THIS process, which the parent ends, instead of the bench."""
    import torch
    import ptr_amd
    torch.cuda.set_device(device)
    stream = torch.cuda.Stream()
    rdr = ptr_amd.Renderer(device)
    rdr.set_scene(example1_soa()); rdr.set_background(np.array([2.0, 2.0, 2.0]))
    rdr.comm_init_rank(bytes.fromhex(uid_hex), rank, world)
    w, h = 512, 256
    part = ptr_amd.StripPart(world, rank, STRIP)
    rows, max_rows = part.local_rows(h), ptr_amd.StripPart(world, 0, STRIP).local_rows(h)
    local = torch.zeros((1, max_rows, w), dtype=torch.int32, device="cuda")
    full = torch.zeros((1, h, w), dtype=torch.int32, device="cuda") if rank == 0 else None
    rdr.render_strips_device(rdr.params(w, h, 0, rows, flags=0), part, 0, 1, local.data_ptr(), stream=stream.cuda_stream,
                             frame_stride_px=max_rows * w)
    rdr.gather_strips_device(w, h, STRIP, 1, local.data_ptr(), full.data_ptr() if rank == 0 else None, root=0,
                             frame_stride_px=max_rows * w, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    good = True
    if rank == 0:
        whole = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        rdr.render_device(rdr.params(w, h, flags=0), 0, 1, whole.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        good = bool(torch.equal(full[0], whole))
    # ... and the other collective the library uses (ncclAllGather, in place, in a group): a small cache's reprojection split over
    # the ranks (the bench checks its result itself later; here it only has to come back)
    w2, h2 = 96, 64
    n2, cells2 = w2 * h2 * 4, (w2 + 2) * (h2 + 2)
    rng = np.random.default_rng(7)
    hits = np.stack([rng.uniform(-30, 30, n2), rng.uniform(-6, 14, n2), rng.uniform(-60, -8, n2)], axis=1)
    rec = np.fromfile(os.path.join(ROOT, "tests", "golden", "rc", "reproject_move_96x64.f64"), count=35)
    cam = rdr.rc_camera(rec[0:3], rec[3:19], rec[19:35].copy(), w2, h2)
    lo, hi = n2 * rank // world, n2 * (rank + 1) // world
    d_hit = torch.from_numpy(hits[lo:hi]).cuda()
    d_age = torch.zeros(hi - lo, dtype=torch.float64, device="cuda")
    d_ip = torch.empty(hi - lo, dtype=torch.int32, device="cuda")
    d_pel = torch.full((cells2,), -1, dtype=torch.int32, device="cuda")
    d_pdep = torch.full((cells2,), 1e6, dtype=torch.float64, device="cuda")
    rdr.rc_reproject_multi_device(cam, hi - lo, lo, d_hit.data_ptr(), d_age.data_ptr(), d_ip.data_ptr(), d_pel.data_ptr(),
                                  d_pdep.data_ptr(), stream=stream.cuda_stream)
    torch.cuda.synchronize()
    print("probe ok" if good else "probe: the gathered frame differs", flush=True)
    os._exit(0 if good else 2)                     # (no teardown: a communicator's destructor is one more collective)


def exchange_probe(uid, rank, world, device, seconds):
    """-> True if this rank's child process finished the probe in time"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--probe", bytes(uid).hex(), str(rank), str(world), str(device)]
    try:
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    except OSError as e:
        print(f"rank {rank}: could not start the exchange probe ({e})", file=sys.stderr)
        return False
    try:
        out, _ = child.communicate(timeout=seconds)
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        print(f"rank {rank}: the exchange probe did not return within {seconds} s (ended)", file=sys.stderr)
        return False
    if child.returncode != 0:
        print(f"rank {rank}: the exchange probe failed (exit {child.returncode}): {out.strip()[-400:]}", file=sys.stderr)
    return child.returncode == 0


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--probe":
        exchange_probe_child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)      # (the chip needs several launches to reach its clocks)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rc", action="store_true", help="skip the render-cache reprojection line (BASELINE configs[4])")
    ap.add_argument("--check", action="store_true", help="after timing, compare rank 0's gathered frame 0 with a single-GPU render")
    ap.add_argument("--mode", choices=["tolerant", "exact"], default="tolerant",
                    help="precision mode of the headline: tolerant = PTR_MODE_TOLERANT (<= 1/255 per channel on every pixel), "
                         "exact = the library's default (bytes equal the reference's)")
    args = ap.parse_args()

    import torch
    import ptr_amd
    if os.environ.get("PTR_BENCH_HANG_DUMP"):            # rehearsals: where is everybody after that many seconds?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PTR_BENCH_HANG_DUMP"]), exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # PTR_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks: ranks share
    # GPUs (local_rank modulo the device count) and the gather is staged through host memory.
    backend = os.environ.get("PTR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    # Everything below runs on ONE explicit stream: the render calls get its handle, torch's own kernels and the
    # RCCL collectives order themselves against torch's current stream.  (The legacy default stream has handle 0,
    # which the C ABI reads as "use the context's own stream" — a non-blocking stream that neither torch nor RCCL
    # would wait for.)
    torch.cuda.set_stream(torch.cuda.Stream())
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    soa = example1_soa()
    bg = np.array([2.0, 2.0, 2.0])
    head_flags = ptr_amd.MODE_TOLERANT if args.mode == "tolerant" else 0
    frames_total = args.frames * world            # weak scaling: fixed per-GPU work
    rdr = ptr_amd.Renderer(local_rank)
    rdr.set_scene(np.ascontiguousarray(np.broadcast_to(soa, (frames_total,) + soa.shape)))
    rdr.set_background(bg)
    stream = torch.cuda.current_stream().cuda_stream
    assert stream != 0

    # ---- per-frame work counts from the GPU's own counting kernel (outside the timed region)
    one = torch.empty(WIDTH * HEIGHT, dtype=torch.int32, device="cuda")
    p_stats = rdr.params(WIDTH, HEIGHT, flags=ptr_amd.FLAG_STATS)
    rdr.render_device(p_stats, 0, 1, one.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    st = rdr.stats()
    rays_per_frame = st["primary"] + st["shadow_rays"]
    flops_per_frame = algorithmic_flops(st)

    # ---- this rank's share: cyclic strips of STRIP rows (all rows when world == 1)
    if world == 1:
        local_rows = HEIGHT
        part = ptr_amd.StripPart(1, 0, HEIGHT)
    else:
        part = ptr_amd.StripPart(world, rank, STRIP)
        local_rows = part.local_rows(HEIGHT)
    max_rows = ptr_amd.StripPart(world, 0, STRIP).local_rows(HEIGHT) if world > 1 else HEIGHT
    n_buf = 2 if world > 1 else 1                # double-buffered: a gather reads one while the next step renders into the other
    locals_ = [torch.empty((frames_total, max_rows, WIDTH), dtype=torch.int32, device="cuda") for _ in range(n_buf)]
    local = locals_[0]
    p = rdr.params(WIDTH, HEIGHT, 0, local_rows, flags=0)
    full = None
    if world > 1 and rank == 0:
        full = torch.empty((frames_total, HEIGHT, WIDTH), dtype=torch.int32, device="cuda")

    # ---- N > 1: the library's communicator (RCCL), checked; torch.distributed is the channel for the id and the fallback
    gather_mode = None
    comm_stream = None
    watchdog = None
    if world > 1:
        # The library's exchange has its first multi-GPU runs in this very start-up phase.  An exception or a wrong frame
        # falls back to torch.distributed below; a collective that never returns cannot be fallen back from, so a
        # watchdog turns it into a quick, explained failure instead of a run that sits until the driver's limit.
        # PTR_BENCH_GATHER=torch skips the library's exchange altogether.
        import threading

        def _stuck():
            print(f"rank {rank}: the start-up phase of the multi-GPU exchange (communicator set-up + first checked gather) "
                  f"did not finish within {WATCHDOG_S} s; set PTR_BENCH_GATHER=torch to run with torch.distributed's gather",
                  file=sys.stderr, flush=True)
            os._exit(3)
        WATCHDOG_S = int(os.environ.get("PTR_BENCH_WATCHDOG_S", "300"))
        watchdog = threading.Timer(WATCHDOG_S, _stuck)
        watchdog.daemon = True
        watchdog.start()
        coll_dev = "cuda" if backend == "nccl" else "cpu"

        def agreed(ok):                          # every rank learns whether every rank is fine
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def shared_id():                         # (always the same collectives on every rank, whatever rank 0 runs into)
            uid = torch.zeros(128, dtype=torch.uint8)
            fine = True
            if rank == 0:
                try:
                    uid = torch.frombuffer(bytearray(rdr.comm_unique_id()), dtype=torch.uint8).clone()
                except Exception as e:           # noqa: BLE001
                    print(f"rank 0: no RCCL id from the library ({e})", file=sys.stderr)
                    fine = False
            uid = uid.to(coll_dev)
            dist.broadcast(uid, src=0)
            return bytes(uid.cpu().numpy().tobytes()) if agreed(fine) else None

        why = None
        if backend != "nccl" and os.environ.get("PTR_BENCH_PROBE") != "force":
            why = "rehearsal backend: ranks may share a GPU, which RCCL refuses"
        elif os.environ.get("PTR_BENCH_GATHER") == "torch":
            why = "PTR_BENCH_GATHER=torch"
        use_library = agreed(why is None)
        if use_library:
            # first in a child process per rank: communicator + one small checked gather.  If that hangs or fails, the child is
            # ended and this process never touches the library's exchange (PTR_BENCH_PROBE=0 skips the probe).
            if os.environ.get("PTR_BENCH_PROBE") != "0":
                pid = shared_id()
                probe_ok = pid is not None and exchange_probe(pid, rank, world, local_rank,
                                                              int(os.environ.get("PTR_BENCH_PROBE_S", "120")))
                use_library = agreed(probe_ok)
                if not use_library:
                    why = "the library's exchange did not pass its probe in a child process"
        if use_library:
            uid = shared_id()
            ok = uid is not None
            if ok:
                try:
                    rdr.comm_init_rank(uid, rank, world)
                except Exception as e:           # noqa: BLE001 - whatever went wrong, every rank has to learn of it
                    ok = False
                    why = str(e)
            use_library = agreed(ok)
        if not use_library:
            print(f"rank {rank}: library communicator not used ({why or 'another rank reported a failure'}); "
                  f"gathering with torch.distributed", file=sys.stderr)
        gather_mode = "library" if use_library else "torch"
        comm_stream = torch.cuda.Stream()
    render_stream = torch.cuda.current_stream()
    rendered = [torch.cuda.Event() for _ in range(n_buf)]
    gathered_ev = [torch.cuda.Event() for _ in range(n_buf)]
    gathered_t = [None]                          # torch fallback only: allocated when that path is first taken (N x a GPU's pixels)
    step_no = [0]

    def gather_step(buf):
        """on the communication stream: the strips of locals_[buf] of every rank -> whole frames on rank 0"""
        if gather_mode == "library":
            rdr.gather_strips_device(WIDTH, HEIGHT, STRIP, frames_total, locals_[buf].data_ptr(),
                                     full.data_ptr() if rank == 0 else None, root=0,
                                     frame_stride_px=max_rows * WIDTH, stream=comm_stream.cuda_stream)
            return
        if rank == 0 and gathered_t[0] is None:
            gathered_t[0] = [torch.empty_like(local) for _ in range(world)]
        if backend == "nccl":
            dist.gather(locals_[buf], gathered_t[0], dst=0)
        else:                                      # rehearsal: stage through the host
            comm_stream.synchronize()
            host = locals_[buf].cpu()
            hl = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, hl, dst=0)
            if rank == 0:
                for r in range(world):
                    gathered_t[0][r].copy_(hl[r])
        if rank == 0:                              # the library's copy kernel puts every part's rows at their places
            for r in range(world):
                rdr.place_strips_device(ptr_amd.StripPart(world, r, STRIP), WIDTH, HEIGHT, frames_total, gathered_t[0][r].data_ptr(),
                                        full.data_ptr(), frame_stride_px=max_rows * WIDTH, stream=comm_stream.cuda_stream)

    def step(flags=None, gather=True):
        p.flags = head_flags if flags is None else flags
        buf = step_no[0] % n_buf
        step_no[0] += 1
        if world > 1:
            render_stream.wait_event(gathered_ev[buf])          # the gather of two steps ago has read this buffer
        rdr.render_strips_device(p, part, 0, frames_total, locals_[buf].data_ptr(), stream=stream,
                                 frame_stride_px=max_rows * WIDTH)
        if world > 1 and gather:
            rendered[buf].record(render_stream)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(rendered[buf])
                gather_step(buf)
                gathered_ev[buf].record(comm_stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- N > 1: before anything is timed, rank 0's first gathered frames must equal single-GPU renders
    if world > 1:
        try:
            step()
            torch.cuda.synchronize()                 # (no torch collective while the library's own is still in flight)
            err = 0
        except ptr_amd.PtrError as e:                # the library's gather refused: every rank must learn of it
            print(f"rank {rank}: {e}", file=sys.stderr)
            err = 1
        flag = torch.tensor([err], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and gather_mode == "library":
            gather_mode = "torch (the library's gather raised an error)"
            step()
        fence()
        good = 1
        if rank == 0:
            whole = torch.empty((HEIGHT, WIDTH), dtype=torch.int32, device="cuda")
            for f in (0, frames_total - 1):
                rdr.render_device(rdr.params(WIDTH, HEIGHT, flags=head_flags), f, 1, whole.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                good &= int(torch.equal(full[f], whole))
        flag = torch.tensor([good], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.broadcast(flag, src=0)
        if int(flag.item()) != 1 and gather_mode == "library":
            if rank == 0:
                print("the library's gather gave a wrong frame; falling back to torch.distributed", file=sys.stderr)
            gather_mode = "torch (the library's gather failed its check)"
            step(); fence()
            if rank == 0:
                good = int(torch.equal(full[0], whole) if frames_total == 1 else torch.equal(full[frames_total - 1], whole))
            flag = torch.tensor([good], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
        if int(flag.item()) != 1:
            raise SystemExit("gathered frame differs from the single-GPU frame")
        watchdog.cancel()

    # The chip reaches its clocks only after some tens of milliseconds of work, and the driver's run (--steps 20 --warmup 5)
    # is shorter than that as a whole: a quarter of a second of the same steps first, untimed like the W warm-up steps
    # that follow, so that the K timed steps measure the steady state (reported as `pre_warm_s`).
    PRE_WARM_S = 0.25
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < PRE_WARM_S:           # (by the clock: every rank for itself, hence without the exchange)
        for _ in range(4):
            step(gather=False)
        torch.cuda.synchronize()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- N > 1: the same steps without the exchange, timed the same way, so that the line shows what the gather costs
    render_only = None
    if world > 1:
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(gather=False)
        fence()
        ro = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(ro, op=dist.ReduceOp.MAX)
        render_only = float(ro.item())

    # ---- dominant kernel's own duration: HIP events on the launch stream, live, same workload
    kms = []
    for _ in range(min(args.steps, 10)):
        step(head_flags | ptr_amd.FLAG_TIMING)
        kms.append(rdr.kernel_ms())
    fence()
    main_ms = float(np.mean([k[0] for k in kms]))
    retrace_ms = float(np.mean([k[1] for k in kms]))

    out = None
    if rank == 0:
        share = local_rows / HEIGHT                      # fraction of every frame this rank renders
        total_rays = rays_per_frame * frames_total       # all ranks together, per step
        value = total_rays * args.steps / elapsed / 1e6
        launch_flops = flops_per_frame * frames_total * share
        launch_bytes = 4.0 * WIDTH * local_rows * frames_total
        achieved = launch_flops / (main_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/sec (primary+secondary) at 1080p",
            "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (discontinuous doubts re-traced in f64; every channel of every pixel within 1/255 of the f64 reference: "
                      "PTR_MODE_TOLERANT)") if head_flags else
                     "f32 (uncertain pixels re-traced in f64; output bytes equal the f64 reference)",
            "timed_region_s": round(elapsed, 4), "pre_warm_s": PRE_WARM_S,
            "data": "synthetic",
            "config": {"workload": f"example1 scene (7 spheres, 2 lights) {WIDTH}x{HEIGHT}, 1 spp, reference "
                                   f"semantics (depth 0), {args.frames} frames per GPU per step, "
                                   + ("tolerant hybrid mode (<= 1/255)" if head_flags else "hybrid mode (reference bytes)"),
                       "frames_per_step": frames_total, "rays_per_frame": rays_per_frame,
                       "parallelism": "1 GPU" if world == 1 else
                       f"{world} ranks, cyclic {STRIP}-row strips, RCCL gather to rank 0 in the timed region, "
                       f"overlapped with the next step's render"},
            "Mpixels_per_s": round(WIDTH * HEIGHT * frames_total * args.steps / elapsed / 1e6, 1),
            "roofline": {"bound": "valu", "kernel": "render_fast_kernel<tolerant>" if head_flags else "render_fast_kernel<hybrid>",
                         "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
                         "frac_step": round(launch_flops / (elapsed / args.steps) / 1e12 / PEAK_FP32_VALU_TFLOPS, 4),
                         "traffic": int(PROFILED_HBM_BYTES_PER_PIXEL["tolerant" if head_flags else "exact"] * WIDTH * local_rows * frames_total),
                         "traffic_source": "not measured in this run: PMC bytes per pixel of the profiled run of this command "
                                           "(profiles/r03_summary.md, WRITE_SIZE + 2 x FETCH_SIZE passes) x pixels of this "
                                           "launch; algorithmic bytes = 4 per pixel",
                         "kernel_ms": round(main_ms, 4), "retrace_kernel_ms": round(retrace_ms, 4),
                         "algorithmic_flops_per_launch": int(launch_flops),
                         "hbm_frac": round(launch_bytes / (main_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                         "note": "FP32 vector-ALU bound (SURVEY.md 8d): no dense contraction, 4 B/pixel of HBM "
                                 "traffic; achieved = algorithmic FLOPs of the reference's control flow / the float32 kernel's "
                                 "own time; frac_step = the same FLOPs / the whole timed step (float32 kernel + float64 "
                                 "re-trace + gaps), i.e. what the path delivers with the reference's bytes"},
        }
        if render_only is not None:
            out["phases"] = {
                "render_only_ms_per_step": round(render_only / args.steps * 1e3, 4),
                "render_only_Mrays_per_s": round(total_rays * args.steps / render_only / 1e6, 1),
                "gather_not_hidden_ms_per_step": round((elapsed - render_only) / args.steps * 1e3, 4),
                "gathered_bytes_per_step_into_rank0": int(4 * WIDTH * HEIGHT * frames_total * (world - 1) / world),
                "gather": gather_mode, "gather_overlapped_with_next_render": True,
                "rank0_inbound_link_bound_ms_per_step": round(4 * WIDTH * HEIGHT * frames_total / world / 64e9 * 1e3, 3),
                "note": "at 1 sample per pixel one GPU produces 0.8 TB/s of pixels; every peer's share of a step crosses ONE xGMI "
                        "link into rank 0 (about 64 GB/s in one direction): the figure above is that link's time for a step, "
                        "which no overlap can hide once it exceeds the render time (DESIGN.md section 6)"}

    # ---- N > 1: what follows (configs[3], the sharded reprojection pass) are more first runs of library collectives.  If one of
    # them never returns, the headline line above is printed as it stands and the process ends, instead of nothing at all.
    extras_watchdog = None
    if world > 1:
        def _extras_stuck():
            if rank == 0:
                out["note_extras"] = ("config3_4k_256spp / config5_rc_reprojection did not finish within "
                                      f"{EXTRAS_S} s and were cut off; the headline figures are complete")
                print(json.dumps(out), flush=True)
            os._exit(0)
        EXTRAS_S = int(os.environ.get("PTR_BENCH_EXTRAS_S", "240"))
        extras_watchdog = threading.Timer(EXTRAS_S + (0 if rank == 0 else 5), _extras_stuck)
        extras_watchdog.daemon = True
        extras_watchdog.start()

    # ---- N = 1: the other precision mode on the same workload, timed the same way; the latency of ONE frame per launch
    other = None
    per_launch = None
    if world == 1:
        oflags = 0 if head_flags else ptr_amd.MODE_TOLERANT
        for _ in range(min(args.warmup, 5)):
            step(oflags)
        fence()
        n_o = max(3, min(args.steps, 50))
        t_o = time.perf_counter()
        for _ in range(n_o):
            step(oflags)
        fence()
        dt_o = time.perf_counter() - t_o
        ko = []
        for _ in range(min(args.steps, 10)):
            step(oflags | ptr_amd.FLAG_TIMING)
            ko.append(rdr.kernel_ms())
        fence()
        other = {"ms_per_step": dt_o / n_o * 1e3, "main_ms": float(np.mean([k[0] for k in ko])),
                 "retrace_ms": float(np.mean([k[1] for k in ko])), "steps": n_o}
        # one 1080p frame per call, host waits for it each time (launch-bound at 1 spp)
        p1 = rdr.params(WIDTH, HEIGHT, flags=head_flags)
        lat = []
        for k in range(60):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rdr.render_device(p1, 0, 1, one.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            if k >= 10:
                lat.append(time.perf_counter() - t1)
        lat_ms = float(np.median(lat)) * 1e3
        per_launch = {"workload": f"one {WIDTH}x{HEIGHT} frame per call (float32 kernel + float64 re-trace), the host waits for every frame",
                      "wall_ms": round(lat_ms, 4), "Mrays_per_s": round(rays_per_frame / lat_ms / 1e3, 1),
                      "frac_of_valu_peak": round(flops_per_frame / (lat_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 4),
                      "calls": len(lat), "statistic": "median"}

    if args.check and world > 1:                      # (a collective: every rank takes the step)
        step(); fence()
    if args.check and rank == 0:
        whole = torch.empty((HEIGHT, WIDTH), dtype=torch.int32, device="cuda")
        rdr.render_device(rdr.params(WIDTH, HEIGHT, flags=head_flags), 0, 1, whole.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        got = full[0] if world > 1 else local[0]
        assert torch.equal(got, whole), "gathered frame differs from the single-GPU frame"
        print("check ok: gathered frame == single-GPU frame", file=sys.stderr)
    # ---- BASELINE.json configs[1] as written ("depth 5"): an EXTENSION (the reference has no bounce loop,
    # SURVEY.md F1), reported beside the headline, same scene / size / batch.  `value` is the mode that meets north_star's
    # tolerance on every pixel (the library's default in the extension modes: the float64 specification, bit for bit); the
    # float32 throughput form (PTR_MODE_F32_ONLY) is its sibling, with the number of pixels it puts beyond 1/255 counted here.
    depth5 = None
    if world == 1:
        p5s = rdr.params(WIDTH, HEIGHT, flags=ptr_amd.FLAG_STATS, max_depth=5)
        rdr.render_device(p5s, 0, 1, one.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        st5 = rdr.stats()
        rays5 = st5["primary"] + st5["secondary"] + st5["shadow_rays"]
        flops5 = algorithmic_flops_ext(st5, 2) * frames_total

        def time_ext(flags, n5):
            p5 = rdr.params(WIDTH, HEIGHT, flags=flags, max_depth=5)
            for _ in range(2):
                rdr.render_device(p5, 0, frames_total, local.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            for _ in range(n5):
                rdr.render_device(p5, 0, frames_total, local.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            dt5 = time.perf_counter() - t5
            p5.flags = flags | ptr_amd.FLAG_TIMING
            k5 = []
            for _ in range(3):
                rdr.render_device(p5, 0, frames_total, local.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                k5.append(rdr.kernel_ms()[0])
            return dt5 / n5 * 1e3, float(np.mean(k5))

        def ext_entry(ms5, k5, peak, kernel, dtype):
            ach5 = flops5 / (k5 * 1e-3) / 1e12
            return {"value": round(rays5 * frames_total / ms5 / 1e3, 1), "unit": "Mrays/s", "ms_per_step": round(ms5, 4), "dtype": dtype,
                    "roofline": {"bound": "valu", "kernel": kernel, "achieved": round(ach5, 3), "peak": peak, "unit": "TFLOP/s",
                                 "frac": round(ach5 / peak, 4), "kernel_ms": round(k5, 4),
                                 "algorithmic_flops_per_launch": int(flops5), "traffic": None}}

        ms64, k64 = time_ext(0, max(3, min(args.steps, 5)))
        ms32, k32 = time_ext(ptr_amd.MODE_F32_ONLY, max(3, min(args.steps, 10)))
        # pixels of one frame that the float32 form puts beyond 1/255 of the specification (compared on the GPU)
        f64_frame = torch.empty(WIDTH * HEIGHT, dtype=torch.int32, device="cuda")
        rdr.render_device(rdr.params(WIDTH, HEIGHT, flags=0, max_depth=5), 0, 1, f64_frame.data_ptr(), stream=stream)
        rdr.render_device(rdr.params(WIDTH, HEIGHT, flags=ptr_amd.MODE_F32_ONLY, max_depth=5), 0, 1, one.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        d8 = (f64_frame.view(torch.uint8).to(torch.int16) - one.view(torch.uint8).to(torch.int16)).abs().reshape(-1, 4).amax(dim=1)
        beyond = int((d8 > 1).sum().item())
        depth5 = {"workload": f"BASELINE.json configs[1]: example1 {WIDTH}x{HEIGHT}, depth 5, 1 spp, {frames_total} frames per step",
                  "mode": "default of the extension modes: the float64 specification bit for bit (0 pixels beyond 1/255 by construction)",
                  **ext_entry(ms64, k64, PEAK_FP32_VALU_TFLOPS / 2, "render_ext_f64_kernel", "f64"),
                  "rays_per_frame": rays5, "camera_rays": st5["primary"], "secondary_rays": st5["secondary"],
                  "shadow_rays": st5["shadow_rays"],
                  "f32_form": {**ext_entry(ms32, k32, PEAK_FP32_VALU_TFLOPS, "render_ext_f32_kernel", "f32"),
                               "pixels_beyond_1_255": beyond, "pixels": WIDTH * HEIGHT,
                               "note": "PTR_MODE_F32_ONLY: the throughput form; it misses north_star's tolerance on the pixels "
                                       "counted here (discontinuities and their mirror images), which is why it is not `value`"},
                  "note": "extension mode, specification in include/ptr.h; not reference behaviour (parity vs the specification: "
                          "tests/test_gpu_ext.py); algorithmic FLOPs of the specification's control flow (bench.py "
                          "algorithmic_flops_ext) / the kernel's own time (HIP events); the float64 roofline is priced against "
                          "the FP64 vector peak (half the FP32 one); HBM traffic: profiles/"}

    # ---- BASELINE.json configs[3]'s workload beside the headline (4K, 256 samples per pixel, depth 5, float32 extension
    # kernel, synthetic.config3_scene): one 4K frame's worth of pixels per GPU per step.  N > 1: `world` frames cut into cyclic
    # strips, every rank renders its part of each, gathered on rank 0.  N = 1: the 8 strip parts of ONE frame rendered one after
    # the other on the one GPU (what each of 8 ranks would render, no gather) - the first point of the 1 -> 8 curve on the
    # workload that can scale: a pixel costs ~10^3 times its 4 bytes here.
    config3 = None
    if True:
        from ptr_amd import synthetic as syn
        W4, H4, SPP, SEED = 3840, 2160, 256, 0x5EED1234
        parts3 = world if world > 1 else 8
        r3 = ptr_amd.Renderer(local_rank)
        r3.set_scene(np.ascontiguousarray(np.broadcast_to(syn.config3_scene(), (world, 12, 7))))
        r3.set_background(syn.BACKGROUND)
        max3 = ptr_amd.StripPart(parts3, 0, STRIP).local_rows(H4)
        my_parts = [ptr_amd.StripPart(parts3, rank, STRIP)] if world > 1 else [ptr_amd.StripPart(parts3, k, STRIP) for k in range(parts3)]
        loc3 = torch.empty((len(my_parts), world, max3, W4), dtype=torch.int32, device="cuda")
        full3 = torch.empty((world, H4, W4), dtype=torch.int32, device="cuda") if (rank == 0 and world > 1) else None
        gat3 = [torch.empty_like(loc3[0]) for _ in range(world)] if (rank == 0 and world > 1 and gather_mode != "library") else None

        def step3():
            for k, part3 in enumerate(my_parts):
                p3 = r3.params(W4, H4, 0, part3.local_rows(H4), flags=ptr_amd.MODE_F32_ONLY, max_depth=5, spp=SPP, seed=SEED)
                r3.render_strips_device(p3, part3, 0, world, loc3[k].data_ptr(), stream=stream, frame_stride_px=max3 * W4)
            if world == 1:
                return
            if gather_mode == "library":
                rdr.gather_strips_device(W4, H4, STRIP, world, loc3[0].data_ptr(), full3.data_ptr() if rank == 0 else None,
                                         root=0, frame_stride_px=max3 * W4, stream=stream)
            else:
                if backend == "nccl":
                    dist.gather(loc3[0], gat3, dst=0)
                else:
                    torch.cuda.synchronize()
                    host = loc3[0].cpu()
                    hl = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                    dist.gather(host, hl, dst=0)
                    if rank == 0:
                        for r in range(world):
                            gat3[r].copy_(hl[r])
                if rank == 0:
                    for r in range(world):
                        rdr.place_strips_device(ptr_amd.StripPart(world, r, STRIP), W4, H4, world, gat3[r].data_ptr(),
                                                full3.data_ptr(), frame_stride_px=max3 * W4, stream=stream)
        for _ in range(2):
            step3()
        fence()
        n3 = max(3, min(args.steps, 10))
        t3 = time.perf_counter()
        for _ in range(n3):
            step3()
        fence()
        d3 = torch.tensor([time.perf_counter() - t3], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(d3, op=dist.ReduceOp.MAX)
        if rank == 0:
            # rays per 4K frame from the work counters of a 1/16-size frame of the same picture (x 16), counted in float64
            r3.render_device(r3.params(W4 // 4, H4 // 4, flags=ptr_amd.FLAG_STATS, max_depth=5, spp=SPP, seed=SEED), 0, 1,
                             loc3.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            c3 = r3.stats()
            rays3 = 16 * (c3["primary"] + c3["secondary"] + c3["shadow_rays"])
            config3 = {"workload": f"BASELINE.json configs[3]: {W4}x{H4}, {SPP} spp, depth 5, {world} frame(s) per step in cyclic "
                                   f"{STRIP}-row strips (one frame's worth of pixels per GPU)"
                                   + (f" over {world} GPUs, gathered on rank 0" if world > 1 else
                                      f": the {parts3} parts of an 8-GPU split rendered one after the other on the one GPU, no gather"),
                       "value": round(rays3 * world * n3 / float(d3.item()) / 1e6, 1), "unit": "Mrays/s", "n_gpus": world, "scaling": "weak",
                       "ms_per_step": round(float(d3.item()) / n3 * 1e3, 3), "rays_per_frame_estimate": rays3,
                       "gathered_bytes_per_step_into_rank0": int(4 * W4 * H4 * world * (world - 1) / world),
                       "dtype": "f32", "note": "extension mode (specification in include/ptr.h), not reference behaviour; float32 "
                                               "throughput form (PSNR >= 50 dB against the specification asserted in tests/test_gpu_ext.py); "
                                               "ray count scaled from a 960x540 counting run"}
        r3.close()

    # ---- BASELINE.json configs[4], second half: the render cache's reprojection pass, the cache split over the ranks
    rc_line = None
    if not args.no_rc:
        rc_line = rc_reprojection_line(ptr_amd, torch, dist, rdr, world, rank, backend if world > 1 else "none",
                                       bool(gather_mode == "library"), stream, max(5, min(args.steps, 20)))

    if rank == 0:
        if other is not None:
            o_val = total_rays / other["ms_per_step"] / 1e3
            o_ach = launch_flops / (other["main_ms"] * 1e-3) / 1e12
            out["bit_exact" if head_flags else "tolerant"] = {
                "mode": "the library's default: output bytes equal the reference's" if head_flags else
                        "PTR_MODE_TOLERANT: every channel of every pixel within 1/255 of the reference",
                "value": round(o_val, 1), "unit": "Mrays/s", "ms_per_step": round(other["ms_per_step"], 4), "steps": other["steps"],
                "roofline": {"bound": "valu", "achieved": round(o_ach, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(o_ach / PEAK_FP32_VALU_TFLOPS, 4),
                             "frac_step": round(launch_flops / (other["ms_per_step"] * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 4),
                             "kernel_ms": round(other["main_ms"], 4), "retrace_kernel_ms": round(other["retrace_ms"], 4)}}
        if per_launch is not None:
            out["per_launch"] = per_launch
        if config3 is not None:
            out["config3_4k_256spp"] = config3
        if depth5 is not None:
            out["extension_depth5"] = depth5
        if rc_line is not None:
            out["config5_rc_reprojection"] = rc_line
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(soa, bg)
            out["cpu_baseline_js"] = cpu_baseline_js()
    if extras_watchdog is not None:
        extras_watchdog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
