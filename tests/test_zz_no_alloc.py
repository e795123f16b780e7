"""ptr_render_device_noalloc through the C ABI: same bytes as ptr_render_device, which is itself checked against the
oracle elsewhere; two calls in a row, the second with a smaller input (the static array keeps the first call's contents);
that it allocates nothing (free device memory unchanged on a context that has never rendered); and the use include/ptr.h
names for it: recorded into a HIP graph and replayed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_py as orc          # noqa: E402  (checker only)
import ptr_amd                   # noqa: E402


def _render(r, entry, w, h, n_frames, flags, stream, **kw):
    import torch
    out = torch.full((n_frames, h, w), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    p = r.params(w, h, flags=flags, **kw)
    entry(p, 0, n_frames, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


def test_noalloc_entry_matches_render_device(scenes):
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    soa = ptr_amd.scene_to_soa(sc["elements"])
    moved = soa.copy(); moved[1, 1] = 2.5; moved[0, 4] = -4.0
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(np.stack([soa, moved])); r.set_background(sc["background"])
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            s = stream.cuda_stream
            # the upload allocates: the allocation-free call refuses to do it ...
            with pytest.raises(ptr_amd.PtrError) as e:
                _render(r, r.render_device_noalloc, 64, 48, 1, 0, s)
            assert e.value.code == ptr_amd.ERR_INVALID and "ptr_sync" in str(e.value)
            r.sync()                                            # ... ptr_sync does
            # first call: two frames of 640x480 (hybrid: fills the queue in the static array), then a smaller one
            for (w, h, nf) in ((640, 480, 2), (161, 67, 1)):
                want = _render(r, r.render_device, w, h, nf, 0, s)
                got = _render(r, r.render_device_noalloc, w, h, nf, 0, s)
                assert np.array_equal(got, want), (w, h, nf)
                for f, sc_f in enumerate((soa, moved)[:nf]):
                    assert np.array_equal(got[f].ravel(), orc.render(sc_f, sc["background"], w, h, flags=0)), (w, h, f)
            # the other modes share the entry point (their stats words live in the same array)
            for flags, kw in ((ptr_amd.MODE_EXACT_F64, {}), (ptr_amd.MODE_F32_ONLY, {}),
                              (ptr_amd.MODE_EXACT_F64, dict(max_depth=3, spp=2, seed=5))):
                want = _render(r, r.render_device, 96, 64, 1, flags, s, **kw)
                got = _render(r, r.render_device_noalloc, 96, 64, 1, flags, s, **kw)
                assert np.array_equal(got, want), (flags, kw)
            # work counters of an allocation-free call
            _render(r, r.render_device_noalloc, 160, 120, 1, ptr_amd.FLAG_STATS, s)
            st = r.stats()
            _, cnt = orc.render(soa, sc["background"], 160, 120, flags=0, counters=True)
            assert st["primary"] == cnt["primary"] and st["shadow_rays"] == cnt["shadow_rays"]
            # beyond the array: a clear error, nothing launched (1080p x 2 frames needs 34.6 MB of queue, the array holds 33.5)
            with pytest.raises(ptr_amd.PtrError) as e:
                _render(r, r.render_device_noalloc, 1920, 1080, 2, 0, s)
            assert e.value.code == ptr_amd.ERR_INVALID and "static array" in str(e.value)
            with pytest.raises(ptr_amd.PtrError):
                _render(r, r.render_device_noalloc, 64, 48, 1, ptr_amd.FLAG_TIMING, s)
            # one whole 1080p frame fits
            want = _render(r, r.render_device, 1920, 1080, 1, 0, s)
            got = _render(r, r.render_device_noalloc, 1920, 1080, 1, 0, s)
            assert np.array_equal(got, want)
    finally:
        r.close()


def test_noalloc_entry_allocates_nothing(scenes):
    """The entry's defining property, on a context that has NEVER rendered: its queue and counter buffers do not exist yet,
    so the ordinary entry would allocate them in this very call (hipMemGetInfo: free bytes before == after)."""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    soa = ptr_amd.scene_to_soa(sc["elements"])
    w, h = 640, 480
    out = torch.zeros((h, w), dtype=torch.int32, device="cuda")          # (torch's allocation happens before the measurement)
    want = orc.render(soa, sc["background"], w, h, flags=0)
    stream = torch.cuda.Stream()
    # the HIP runtime's own lazy allocations (a stream's hardware queue, a kernel's code on first launch) are made to happen
    # first, through ANOTHER context: what is measured below is this library's behaviour on a context that never rendered
    warm = ptr_amd.Renderer(0)
    warm.set_scene(soa); warm.set_background(sc["background"])
    for flags in (0, ptr_amd.MODE_TOLERANT):
        warm.render_device(warm.params(w, h, flags=flags), 0, 1, out.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    warm.close()
    for flags in (0, ptr_amd.MODE_TOLERANT):
        r = ptr_amd.Renderer(0)
        try:
            r.set_scene(soa); r.set_background(sc["background"])
            r.sync()                                                      # the upload (which allocates) is done here
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            for _ in range(2):
                r.render_device_noalloc(r.params(w, h, flags=flags), 0, 1, out.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            free1 = torch.cuda.mem_get_info()[0]
            assert free1 == free0, (flags, free0 - free1)
            got = out.cpu().numpy().view(np.uint32).ravel()
            if flags == 0:
                assert np.array_equal(got, want)
            else:                                                         # tolerance mode: 1/255 per channel
                assert int(np.abs(got.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16)).max()) <= 1
            # a refused call leaves what ptr_get_stats reads as the last successful call left it
            before = r.stats()
            with pytest.raises(ptr_amd.PtrError):
                r.render_device_noalloc(r.params(4000, 3000, flags=flags), 0, 1, out.data_ptr(), stream=stream.cuda_stream)    # refused: nothing runs
            assert r.stats() == before
            # the ordinary entry on the same context does allocate (the measurement can see an allocation)
            r.render_device(r.params(w, h, flags=flags), 0, 1, out.data_ptr(), stream=stream.cuda_stream)
            stream.synchronize()
            assert torch.cuda.mem_get_info()[0] < free0
        finally:
            r.close()


def test_noalloc_entry_under_graph_capture(scenes):
    """Recorded into a HIP graph (torch.cuda.graph: capture mode "global", under which an allocating HIP call made by any thread
    fails the capture) and replayed twice; afterwards ptr_get_stats and a call on another stream work - the context ties no
    event to the capturing stream."""
    torch = pytest.importorskip("torch")
    sc = scenes["example1"]
    soa = ptr_amd.scene_to_soa(sc["elements"])
    w, h = 320, 240
    want = orc.render(soa, sc["background"], w, h, flags=0)
    r = ptr_amd.Renderer(0)
    try:
        r.set_scene(soa); r.set_background(sc["background"])
        r.sync()
        out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        other = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        p = r.params(w, h, flags=0)
        side = torch.cuda.Stream()
        # an ordinary call first, so that the context holds a "last call" event from another stream when the capture starts
        r.render_device(p, 0, 1, other.data_ptr(), stream=side.cuda_stream)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.graph(g, stream=cap):
            r.render_device_noalloc(p, 0, 1, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        assert not out.any()                                   # nothing ran during the capture
        for _ in range(2):
            out.fill_(0x5A5A5A5A)
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32).ravel(), want)
        st = r.stats()                                         # (hipEventSynchronize on a captured event would fail here)
        assert st["retraced"] + st["reshaded"] > 0
        other.zero_()
        r.render_device(p, 0, 1, other.data_ptr(), stream=side.cuda_stream)
        side.synchronize()
        assert np.array_equal(other.cpu().numpy().view(np.uint32).ravel(), want)
    finally:
        r.close()
