"""Soak of the render cache's two per-frame passes on random inputs: ptr_rc_reproject and
ptr_rc_shade through the C ABI against the CPU restatement (oracle/), every output bit for bit.
    python tools/soak_rc.py [TRIALS] [SEED]          (PTR_SOAK_RANKS=6: also the cache split over 1 .. 6 ranks, ptr_rc_reproject_shard_device
                                                      + ptr_rc_merge_shards_device, the ranks played one after the other on the one GPU)
Each trial: a random grid (8..480 x 8..320), 1..4 items per pixel, a random look-at camera (fov 20..110 degrees, any aspect),
hits scattered in front of, beside and behind the camera with holes (NaN), duplicates (equal depths: the first index wins),
hits at the camera itself and far beyond the depth threshold; then the reprojection's grid goes through depth culling, gap
filling and the three frames with random item colours and ages.  Test infrastructure: nothing here is the product path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py"), os.path.join(ROOT, "oracle"), ROOT]
import numpy as np
import oracle_py as orc
import ptr_amd


def look_at(eye, to, fov, aspect, near=0.1, far=10000.0):
    """column-major matrixWorldInverse and projectionMatrix of a three.js PerspectiveCamera at eye looking at to"""
    z = eye - to; z /= np.linalg.norm(z)
    x = np.cross([0.0, 1.0, 0.0], z)
    x = x / np.linalg.norm(x) if np.linalg.norm(x) > 1e-9 else np.array([1.0, 0.0, 0.0])
    y = np.cross(z, x)
    view = np.array([x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0, -x @ eye, -y @ eye, -z @ eye, 1.0])
    top = near * np.tan(np.pi / 180 * 0.5 * fov); right = top * aspect
    proj = np.array([near / right, 0, 0, 0, 0, near / top, 0, 0, 0, 0, -(far + near) / (far - near), -1,
                     0, 0, -2 * far * near / (far - near), 0])
    return view, proj


def sharded(rs, rng, eye, view, proj, w, h, hits, age, age_factor, threshold, want, inner):
    """the multi-GPU form with its ranks played one after the other on one GPU (a context per rank, the all-gather done by
    the shared arrays): random cuts of the cache in cache order, empty shards among them; every rank's merge must leave the
    whole-cache result for its own items and for the grid"""
    import torch
    n, ranks = hits.shape[0], len(rs)
    cells = (w + 2) * (h + 2)
    cuts = np.sort(rng.integers(0, n + 1, ranks - 1)) if ranks > 1 else np.zeros(0, np.int64)
    if ranks > 2 and rng.random() < 0.3:
        cuts[1] = cuts[0]                                          # an empty shard in the middle
    bounds = list(zip([0, *cuts.tolist()], [*cuts.tolist(), n]))
    cam = ptr_amd.Renderer.rc_camera(eye, view, proj, w, h)
    all_el = torch.full((ranks, cells), -1, dtype=torch.int32, device="cuda")
    all_dep = torch.full((ranks, cells), threshold, dtype=torch.float64, device="cuda")
    dev = []
    for r, (lo, hi) in enumerate(bounds):
        m = max(hi - lo, 1)
        d_hit = torch.zeros((m, 3), dtype=torch.float64, device="cuda"); d_age = torch.zeros(m, dtype=torch.float64, device="cuda")
        d_hit[:hi - lo] = torch.from_numpy(hits[lo:hi]).cuda(); d_age[:hi - lo] = torch.from_numpy(age[lo:hi]).cuda()
        d_ip = torch.full((m,), -7, dtype=torch.int32, device="cuda")
        dev.append((d_hit, d_age, d_ip))
        torch.cuda.synchronize()
        rs[r].rc_reproject_shard_device(cam, hi - lo, lo, d_hit.data_ptr(), d_age.data_ptr(), d_ip.data_ptr(), all_el[r].data_ptr(),
                                        all_dep[r].data_ptr(), age_factor, threshold)
        rs[r].sync()
    bad = 0
    for r, (lo, hi) in enumerate(bounds):
        d_hit, d_age, d_ip = dev[r]
        pel = torch.full((cells,), -1, dtype=torch.int32, device="cuda")
        pdep = torch.full((cells,), threshold, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        rs[r].rc_merge_shards_device(w, h, ranks, r, all_el.data_ptr(), all_dep.data_ptr(), hi - lo, lo, d_age.data_ptr(), d_ip.data_ptr(),
                                     pel.data_ptr(), pdep.data_ptr(), age_factor, threshold)
        rs[r].sync()
        bad += int((d_age[:hi - lo].cpu().numpy() != want[0][lo:hi]).sum()) + int((d_ip[:hi - lo].cpu().numpy() != want[1][lo:hi]).sum())
        bad += int((pel.cpu().numpy()[inner] != want[2][inner]).sum())
        bad += int((pdep.cpu().numpy()[inner].view(np.uint64) != want[3][inner].view(np.uint64)).sum())
    return bad


def trial(r, rng, rs=None):
    w, h = int(rng.integers(8, 481)), int(rng.integers(8, 321))
    n = w * h * int(rng.integers(1, 5))
    eye = rng.uniform(-20, 20, 3)
    to = eye + rng.normal(size=3) * 10
    fov = rng.uniform(20, 110)
    aspect = w / h if rng.random() < 0.7 else rng.uniform(0.3, 3.0)
    view, proj = look_at(eye, to, fov, aspect)
    fwd = (to - eye) / np.linalg.norm(to - eye)
    kind = rng.random(n)
    hits = eye + fwd * rng.uniform(1, 80, (n, 1)) + rng.normal(size=(n, 3)) * rng.uniform(2, 40)        # mostly in view
    behind = kind < 0.1
    hits[behind] = eye - fwd * rng.uniform(0.5, 50, (int(behind.sum()), 1)) + rng.normal(size=(int(behind.sum()), 3))
    hits[(kind >= 0.1) & (kind < 0.25), 0] = np.nan                                                     # no hit
    hits[(kind >= 0.25) & (kind < 0.26)] = eye                                                          # depth 0
    far = (kind >= 0.26) & (kind < 0.28)
    hits[far] = eye + fwd * rng.uniform(1e5, 1e8, (int(far.sum()), 1))
    dup = rng.integers(0, n, n // 50)
    hits[dup] = hits[(dup + rng.integers(1, 200, dup.size)) % n]                                        # equal depths, same pixel
    age = rng.integers(0, 40, n).astype(np.float64)
    age_factor = float(rng.choice([1.0, 0.5, 2.0, 1.25]))
    threshold = float(rng.choice([1e6, 60.0, 1e3]))
    want = orc.rc_reproject(eye, view, proj, w, h, hits, age, age_factor, threshold)
    got = r.rc_reproject(eye, view, proj, w, h, hits, age, age_factor, threshold)
    inner = np.zeros((h + 2, w + 2), dtype=bool); inner[1:h + 1, 1:w + 1] = True
    inner = inner.ravel()
    bad = 0
    bad += int((got[0] != want[0]).sum()) + int((got[1] != want[1]).sum())
    bad += int((got[2][inner] != want[2][inner]).sum())
    bad += int((got[3][inner].view(np.uint64) != want[3][inner].view(np.uint64)).sum())
    if rs:
        bad += sharded(rs[:int(rng.integers(1, len(rs) + 1))], rng, eye, view, proj, w, h, hits, age, age_factor, threshold, want, inner)
    # the second pass on the first one's grid (the border as Pixel.js leaves it: no element, depth 1e13)
    el = np.where(inner, want[2], -1).astype(np.int32)
    dep = np.where(inner, want[3], 10000000000000.0)
    rgb = rng.integers(0, 256, (n, 3)).astype(np.float64)
    zero, rnd = float(rng.choice([20.0, 5.0, 60.0])), float(rng.choice([5.0, 0.0, 17.0]))
    ws = orc.rc_shade(w, h, el, dep, want[0], rgb, age_factor, zero, rnd)
    gs = r.rc_shade(w, h, el, dep, want[0], rgb, age_factor, zero, rnd)
    bad += int((gs["age"] != ws["age"]).sum())
    bad += int((gs["pixel_rgb"][inner] != ws["pixel_rgb"][inner]).sum()) + int((gs["pixel_priority"][inner] != ws["pixel_priority"][inner]).sum())
    bad += int((gs["hist"] != ws["hist"]).sum()) + int((gs["stats"] != ws["stats"]).sum())
    for key in ("color", "reprojection", "priority"):
        bad += int((gs[key] != ws[key]).sum())
    return n, w * h, int((want[1] >= 0).sum()), bad


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    r = ptr_amd.Renderer(0)
    ranks = int(os.environ.get("PTR_SOAK_RANKS", "0"))          # > 0: also the sharded form with 1 .. RANKS ranks played on this GPU
    rs = [ptr_amd.Renderer(0) for _ in range(ranks)]
    items = pixels = winners = bad = 0
    t0 = time.time()
    try:
        for t in range(trials):
            n, px, win, b = trial(r, rng, rs)
            items += n; pixels += px; winners += win; bad += b
            if b:
                print(f"trial {t} (seed {seed}): {b} values differ", flush=True)
            if (t + 1) % 20 == 0:
                print(f"  {t + 1} trials, {items / 1e6:.1f} M items, {time.time() - t0:.0f} s", flush=True)
    finally:
        r.close()
        for x in rs:
            x.close()
    print(f"soak_rc seed {seed}{f', sharded over 1..{ranks} ranks too' if ranks else ''}: {trials} trials, {items / 1e6:.1f} M items over {pixels / 1e6:.1f} M pixels, "
          f"{winners / 1e6:.1f} M pixel winners, {bad} values differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
