"""ptr_render_device_noalloc through the C ABI: same bytes as ptr_render_device, which is itself checked against the
oracle elsewhere; two calls in a row, the second with a smaller input (the static array keeps the first call's contents)."""
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
