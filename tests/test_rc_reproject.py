"""the render cache's reprojection pass (Driver.reprojectFrame + Camera.reprojectPixel over
three.js Vector3.project).  tests/golden/rc/reproject_*.f64 were produced by the UNMODIFIED reference under Node
with the three.js it vendors (oracle/gen_golden_rc_reproject.mjs); the oracle, and the GPU entry point
ptr_rc_reproject through the C ABI, must reproduce every output bit for bit."""
import json
import os

import numpy as np
import pytest

import oracle_py as orc


def load_cases(golden_dir):
    with open(os.path.join(golden_dir, "rc", "reproject_index.json")) as f:
        index = json.load(f)
    for name, m in index["cases"].items():
        rec = np.fromfile(os.path.join(golden_dir, "rc", m["file"]))
        w, h, n = m["width"], m["height"], m["items"]
        cells = (w + 2) * (h + 2)
        assert rec.size == 35 + 4 * n + 2 * n + 2 * cells
        o = 35
        items = rec[o:o + 4 * n].reshape(n, 4); o += 4 * n
        out_items = rec[o:o + 2 * n].reshape(n, 2); o += 2 * n
        pix = rec[o:].reshape(cells, 2)
        yield name, dict(m, cam_from=rec[0:3], view=rec[3:19], proj=rec[19:35], hits=items[:, 0:3].copy(), age_in=items[:, 3].copy(),
                         age_out=out_items[:, 0], item_pixel=out_items[:, 1].astype(np.int32),
                         pixel_element=pix[:, 0].astype(np.int32), pixel_depth=pix[:, 1])


def check(name, c, age, item_pixel, el, dep):
    w, h = c["width"], c["height"]
    inner = np.zeros((h + 2, w + 2), dtype=bool); inner[1:h + 1, 1:w + 1] = True
    inner = inner.ravel()
    assert np.array_equal(age, c["age_out"]), name
    assert np.array_equal(item_pixel, c["item_pixel"]), name
    assert np.array_equal(el[inner], c["pixel_element"][inner]), name
    assert np.array_equal(dep[inner], c["pixel_depth"][inner]), name


def test_oracle_reproduces_the_reference_pass(golden_dir):
    seen = 0
    for name, c in load_cases(golden_dir):
        out = orc.rc_reproject(c["cam_from"], c["view"], c["proj"], c["width"], c["height"], c["hits"], c["age_in"],
                               c["age_factor"], c["depth_threshold"])
        check(name, c, *out)
        # the fixture exercises every branch: no hit, depth 0 (+4), outside (+2), loser (+1), displaced winner, winner
        d = (c["age_out"] - c["age_in"]) / c["age_factor"]
        assert set(np.unique(d)) >= ({0.0, 1.0, 4.0} if name.startswith("same") else {0.0, 1.0, 2.0, 4.0}), name
        displaced = (c["item_pixel"] >= 0) & (c["pixel_element"][np.maximum(c["item_pixel"], 0)] != np.arange(c["items"]))
        assert displaced.any(), name
        seen += 1
    assert seen == 4


@pytest.mark.gpu
def test_gpu_reprojection_bit_exact(golden_dir):
    import ptr_amd
    r = ptr_amd.Renderer(0)
    try:
        for name, c in load_cases(golden_dir):
            out = r.rc_reproject(c["cam_from"], c["view"], c["proj"], c["width"], c["height"], c["hits"], c["age_in"],
                                 c["age_factor"], c["depth_threshold"])
            check(name, c, *out)
    finally:
        r.close()


@pytest.mark.gpu
def test_gpu_reprojection_full_hd_against_oracle():
    """1080p cache (4 items per pixel, 8.3 M items) of synthetic hits: the GPU pass equals the sequential oracle."""
    import ptr_amd
    w, h = 1920, 1080
    n = w * h * 4
    rng = np.random.default_rng(7)
    hits = np.stack([rng.uniform(-30, 30, n), rng.uniform(-6, 14, n), rng.uniform(-60, -8, n)], axis=1)
    hits[rng.random(n) < 0.2, 0] = np.nan
    hits[::1000] = hits[1::1000]                       # equal depths: first index must win
    age = rng.integers(0, 30, n).astype(np.float64)
    name, c = next(load_cases(os.path.join(os.path.dirname(__file__), "golden")))
    proj = c["proj"].copy(); proj[0] = c["proj"][5] / (w / h)       # same fov, 16:9
    want = orc.rc_reproject(c["cam_from"], c["view"], proj, w, h, hits, age)
    r = ptr_amd.Renderer(0)
    try:
        got = r.rc_reproject(c["cam_from"], c["view"], proj, w, h, hits, age)
    finally:
        r.close()
    inner = np.zeros((h + 2, w + 2), dtype=bool); inner[1:h + 1, 1:w + 1] = True
    inner = inner.ravel()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2][inner], want[2][inner]) and np.array_equal(got[3][inner], want[3][inner])
    assert (want[1] >= 0).sum() > 100000


# ---- the cache split over several GPUs in cache order (ptr_rc_reproject_shard_device / _merge_shards_device) ---------
def shard_bounds(n, ranks):
    """uneven on purpose: the first rank gets no item at all when there are three or more"""
    cuts = [0] + [0 if ranks >= 3 and r == 0 else (n * (r + 1)) // ranks + (7 if r + 1 < ranks else 0) for r in range(ranks)]
    cuts[-1] = n
    return [(min(cuts[r], n), min(cuts[r + 1], n)) for r in range(ranks)]


def merge_shards_model(rank, base, local, all_el, all_dep, item_depth, age_factor, depth_threshold):
    """What ptr_rc_merge_shards_device does for one rank, in numpy: local = (age, item_pixel) of the rank's local pass."""
    age, item_pixel = local[0].copy(), local[1].copy()
    cells = all_el.shape[1]
    best = np.full(cells, float(depth_threshold)); before = best.copy()
    element = np.full(cells, -1, np.int32); winner = np.full(cells, -1)
    for r in range(all_el.shape[0]):
        if r == rank:
            before = best.copy()
        take = (all_el[r] >= 0) & (all_dep[r] < best)
        best[take] = all_dep[r][take]; element[take] = all_el[r][take]; winner[take] = r
    lost = (all_el[rank] >= 0) & (winner != rank)              # a local winner that another rank's item beats: aged once
    age[all_el[rank][lost] - base] += age_factor
    has = item_pixel >= 0                                       # pointers of items that an earlier rank's item hides
    hidden = has.copy()
    hidden[has] = ~(item_depth[has] < before[item_pixel[has]])
    item_pixel[hidden] = -1
    return age, item_pixel, element, best


def local_pass_oracle(c, lo, hi):
    """one rank's local pass by the oracle: element indices in the whole cache, plus the items' depths"""
    age, ip, el, dep = orc.rc_reproject(c["cam_from"], c["view"], c["proj"], c["width"], c["height"], c["hits"][lo:hi].copy(),
                                        c["age_in"][lo:hi].copy(), c["age_factor"], c["depth_threshold"])
    el = np.where(el >= 0, el + lo, -1).astype(np.int32)
    d = c["hits"][lo:hi] - c["cam_from"]
    return age, ip, el, dep, np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])


@pytest.mark.parametrize("ranks", [1, 2, 3, 5])
def test_sharded_reprojection_model_equals_the_sequential_walk(golden_dir, ranks):
    """the exchange + merge rule of the multi-GPU pass, on the CPU: shards through the oracle, merged, equal the reference's
    sequential walk over the whole cache (goldens) bit for bit"""
    for name, c in load_cases(golden_dir):
        bounds = shard_bounds(c["items"], ranks)
        locs = [local_pass_oracle(c, lo, hi) for lo, hi in bounds]
        all_el = np.stack([l[2] for l in locs]); all_dep = np.stack([l[3] for l in locs])
        ages, pixels = [], []
        for r, (lo, hi) in enumerate(bounds):
            age, ip, el, dep = merge_shards_model(r, lo, locs[r][:2], all_el, all_dep, locs[r][4], c["age_factor"], c["depth_threshold"])
            ages.append(age); pixels.append(ip)
        check(name, c, np.concatenate(ages), np.concatenate(pixels), el, dep)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 2, 3, 8])
def test_gpu_sharded_reprojection_equals_single_pass(golden_dir, ranks):
    """the multi-GPU pass with its ranks played one after the other on one GPU (a context per rank, the gather done by
    copies): local pass + merge through the C ABI give the reference's result for the whole cache on the goldens, and the
    single-GPU pass's on a 640 x 360 cache of random points"""
    import torch
    import ptr_amd
    cases = list(load_cases(golden_dir))
    w, h = 640, 360
    n = w * h * 4
    rng = np.random.default_rng(11)
    hits = np.stack([rng.uniform(-30, 30, n), rng.uniform(-6, 14, n), rng.uniform(-60, -8, n)], axis=1)
    hits[rng.random(n) < 0.2, 0] = np.nan
    hits[::500] = hits[1::500]
    c0 = cases[0][1]
    proj = c0["proj"].copy(); proj[0] = c0["proj"][5] / (w / h)
    big = dict(c0, width=w, height=h, items=n, hits=hits, age_in=rng.integers(0, 30, n).astype(np.float64), proj=proj)
    rs = [ptr_amd.Renderer(0) for _ in range(ranks)]
    try:
        want = rs[0].rc_reproject(big["cam_from"], big["view"], proj, w, h, hits, big["age_in"], big["age_factor"], big["depth_threshold"])
        big.update(age_out=want[0], item_pixel=want[1], pixel_element=want[2], pixel_depth=want[3])
        for name, c in cases + [("random_640x360", big)]:
            cw, ch = c["width"], c["height"]
            cells = (cw + 2) * (ch + 2)
            cam = ptr_amd.Renderer.rc_camera(c["cam_from"], c["view"], c["proj"], cw, ch)
            bounds = shard_bounds(c["items"], ranks)
            all_el = torch.full((ranks, cells), -1, dtype=torch.int32, device="cuda")
            all_dep = torch.full((ranks, cells), float(c["depth_threshold"]), dtype=torch.float64, device="cuda")
            dev = []
            for r, (lo, hi) in enumerate(bounds):
                m = max(hi - lo, 1)
                d_hit = torch.zeros((m, 3), dtype=torch.float64, device="cuda"); d_age = torch.zeros(m, dtype=torch.float64, device="cuda")
                d_hit[:hi - lo] = torch.from_numpy(c["hits"][lo:hi]).cuda(); d_age[:hi - lo] = torch.from_numpy(c["age_in"][lo:hi]).cuda()
                d_ip = torch.full((m,), -7, dtype=torch.int32, device="cuda")
                dev.append((d_hit, d_age, d_ip))
                torch.cuda.synchronize()
                rs[r].rc_reproject_shard_device(cam, hi - lo, lo, d_hit.data_ptr(), d_age.data_ptr(), d_ip.data_ptr(), all_el[r].data_ptr(),
                                                all_dep[r].data_ptr(), c["age_factor"], c["depth_threshold"])
                rs[r].sync()
            ages, pixels = [], []
            for r, (lo, hi) in enumerate(bounds):
                d_hit, d_age, d_ip = dev[r]
                pel = torch.full((cells,), -1, dtype=torch.int32, device="cuda")
                pdep = torch.full((cells,), float(c["depth_threshold"]), dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                rs[r].rc_merge_shards_device(cw, ch, ranks, r, all_el.data_ptr(), all_dep.data_ptr(), hi - lo, lo, d_age.data_ptr(),
                                             d_ip.data_ptr(), pel.data_ptr(), pdep.data_ptr(), c["age_factor"], c["depth_threshold"])
                rs[r].sync()
                ages.append(d_age[:hi - lo].cpu().numpy()); pixels.append(d_ip[:hi - lo].cpu().numpy())
                check(f"{name} rank {r}/{ranks}", c, np.concatenate(ages + [c["age_out"][hi:]]), np.concatenate(pixels + [c["item_pixel"][hi:]]),
                      pel.cpu().numpy(), pdep.cpu().numpy())
        # one rank, through its communicator: ptr_rc_reproject_multi_device is the plain pass
        if ranks == 1:
            name, c = cases[0]
            cam = ptr_amd.Renderer.rc_camera(c["cam_from"], c["view"], c["proj"], c["width"], c["height"])
            cells = (c["width"] + 2) * (c["height"] + 2)
            rs[0].comm_init_rank(None, 0, 1)
            d_hit = torch.from_numpy(c["hits"]).cuda(); d_age = torch.from_numpy(c["age_in"]).cuda()
            d_ip = torch.zeros(c["items"], dtype=torch.int32, device="cuda")
            pel = torch.full((cells,), -1, dtype=torch.int32, device="cuda")
            pdep = torch.full((cells,), float(c["depth_threshold"]), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            rs[0].rc_reproject_multi_device(cam, c["items"], 0, d_hit.data_ptr(), d_age.data_ptr(), d_ip.data_ptr(), pel.data_ptr(),
                                            pdep.data_ptr(), c["age_factor"], c["depth_threshold"])
            rs[0].sync()
            check(name + " multi/1", c, d_age.cpu().numpy(), d_ip.cpu().numpy(), pel.cpu().numpy(), pdep.cpu().numpy())
    finally:
        for r in rs:
            r.close()
