"""N > 1 path on CPU (gloo, world_size 2 and 3): the cyclic-strip partition that bench.py and
ptr_render_strips_device use, and the gather + re-interleave to rank 0.  Each rank fills its strips
from the CPU oracle (the GPU kernels are covered by the gpu tests), so this checks the host logic:
ptr_strip_rows / StripPart agree, every row is owned by exactly one rank, and the gathered frame
equals the single-rank frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, strip, w, h, ret):
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "path-tracer_amd", "py"), ROOT]
    import oracle_py as orc
    import ptr_amd
    from bench import example1_soa
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    soa, bg = example1_soa(), [2.0, 2.0, 2.0]
    part = ptr_amd.StripPart(world, rank, strip)
    rows = part.global_rows(h)
    assert part.local_rows(h) == len(rows)                    # C ABI (ptr_strip_rows) == Python mirror
    max_rows = ptr_amd.StripPart(world, 0, strip).local_rows(h)
    local = torch.zeros((max_rows, w), dtype=torch.int32)
    # what ptr_render_strips_device writes: this rank's rows, compact, in increasing y
    for k, y in enumerate(rows):
        local[k] = torch.from_numpy(orc.render(soa, bg, w, h, y, 1, flags=0).view(np.int32))
    gathered = [torch.zeros_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, gathered, dst=0)
    if rank == 0:
        full = torch.zeros((h, w), dtype=torch.int32)
        seen = np.zeros(h, dtype=np.int32)
        for r in range(world):
            sp = ptr_amd.StripPart(world, r, strip)
            idx = sp.global_rows(h)
            # the row rule the library's own gather places by (ptr_strip_image_row) == the Python mirror
            assert [sp.image_row(k) for k in range(len(idx))] == idx
            seen[idx] += 1
            full.index_copy_(0, torch.as_tensor(idx), gathered[r][: len(idx)])
        want = orc.render(soa, bg, w, h, flags=0).reshape(h, w)
        ret["ok"] = bool((seen == 1).all()) and np.array_equal(full.numpy().view(np.uint32), want)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,strip,h", [(2, 8, 45), (3, 4, 50)])
def test_strip_partition_and_gather_gloo(world, strip, h):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29650 + world * 7 + strip
    mp.spawn(_worker, args=(world, port, strip, 64, h, ret), nprocs=world, join=True)
    assert ret.get("ok") is True


def _reproject_worker(rank, world, port, ret):
    """configs[4], the render cache's reprojection with the cache split over ranks: every rank runs the pass on its
    items (the CPU oracle here, ptr_rc_reproject_shard_device on a GPU), the local grids are all-gathered (gloo here,
    ncclAllGather in ptr_rc_reproject_multi_device), and the merge rule of ptr_rc_merge_shards_device (its numpy model)
    gives every rank its part of the reference's sequential result."""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "path-tracer_amd", "py"), os.path.join(ROOT, "tests"), ROOT]
    import test_rc_reproject as t
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for name, c in t.load_cases(os.path.join(ROOT, "tests", "golden")):
        lo, hi = t.shard_bounds(c["items"], world)[rank]
        age, ip, el, dep, item_depth = t.local_pass_oracle(c, lo, hi)
        inner = np.zeros((c["height"] + 2, c["width"] + 2), dtype=bool); inner[1:-1, 1:-1] = True
        dep = np.where(inner.ravel(), dep, float(c["depth_threshold"]))          # (the pass leaves the border alone)
        el_t, dep_t = torch.from_numpy(el.copy()), torch.from_numpy(dep)
        all_el = [torch.zeros_like(el_t) for _ in range(world)]
        all_dep = [torch.zeros_like(dep_t) for _ in range(world)]
        dist.all_gather(all_el, el_t)
        dist.all_gather(all_dep, dep_t)
        age, ip, pel, pdep = t.merge_shards_model(rank, lo, (age, ip), torch.stack(all_el).numpy(), torch.stack(all_dep).numpy(),
                                                  item_depth, c["age_factor"], c["depth_threshold"])
        # every rank holds the whole grid; its items are its slice of the reference's arrays
        ok = ok and np.array_equal(age, c["age_out"][lo:hi]) and np.array_equal(ip, c["item_pixel"][lo:hi])
        ok = ok and np.array_equal(pel[inner.ravel()], c["pixel_element"][inner.ravel()])
        ok = ok and np.array_equal(pdep[inner.ravel()], c["pixel_depth"][inner.ravel()])
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        ret["ok"] = all(flags)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_reprojection_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reproject_worker, args=(world, 29720 + world, ret), nprocs=world, join=True)
    assert ret.get("ok") is True


def test_gather_sizes_at_eight_ranks_and_headline_batch():
    """The strip gather's host arithmetic (ptr_gather_plan = what ptr_gather_strips_device / _all compute) at sizes no test
    box holds: bench.py's headline at N = 8 (2 048 frames of 1920 x 1080 per step, 8-row strips: parts of 136 / 128 rows) and
    BASELINE configs[3] (8 frames of 3840 x 2160).  Everything is checked against Python's unbounded integers: a 32-bit
    product anywhere in the chain (count = stride x frames, the root's staging offsets, the copy kernel's pixel index)
    would show as a mismatch here before the first real 8-GPU run."""
    sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
    import ptr_amd
    for (w, h, frames, n) in ((1920, 1080, 256 * 8, 8), (3840, 2160, 8, 8), (1920, 1080, 256 * 2, 2), (1920, 1080, 65535, 8)):
        strips = -(-h // 8)
        rows_of = [sum(min(8, h - 8 * k) for k in range(r, strips, n)) for r in range(n)]
        total_stage = 0
        for r in range(n):
            g = ptr_amd.gather_plan(w, h, 8, frames, r, n)
            assert g["rows"] == rows_of[r] == ptr_amd.StripPart(n, r, 8).local_rows(h)
            assert g["rows_max"] == rows_of[0] == max(rows_of)
            assert g["frame_stride_px"] == rows_of[0] * w
            assert g["count"] == rows_of[0] * w * frames
            assert g["local_bytes"] == 4 * rows_of[0] * w * frames
            assert g["root_stage_bytes"] == (n - 1) * 4 * rows_of[0] * w * frames
            assert g["full_bytes"] == 4 * w * h * frames
            last_row = ptr_amd.StripPart(n, r, 8).global_rows(h)[-1]
            assert g["last_full_index"] == ((frames - 1) * h + last_row) * w + (w - 1)
            assert g["last_full_index"] < w * h * frames
            total_stage += g["local_bytes"]
        if (w, h, frames, n) == (1920, 1080, 2048, 8):
            assert rows_of == [136, 136, 136, 136, 136, 136, 136, 128]
            assert g["count"] == 534_773_760 and g["root_stage_bytes"] == 14_973_665_280      # > 2^32: the point of this test
            assert g["full_bytes"] == 16_986_931_200 and g["last_full_index"] > 2 ** 31      # (a signed 32-bit index would wrap)
        if frames == 65535:
            assert g["last_full_index"] > 2 ** 32 and g["count"] > 2 ** 32
    # arguments that describe no gather are refused, not wrapped
    for bad in ((0, 1080, 8, 1, 0, 8), (1920, 1080, 8, 70000, 0, 8), (1920, 1080, 8, 1, 8, 8), (1920, 1080, 8, 1, 0, 65), (1 << 24, 8, 8, 1, 0, 2)):
        with pytest.raises(ptr_amd.PtrError):
            ptr_amd.gather_plan(*bad)
