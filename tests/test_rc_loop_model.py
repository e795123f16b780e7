"""The render cache's frame loop on the GPU (ptr_kernels_rcloop.hip) replaces three sequential walks of the reference by
parallel forms: the error diffusion with its stale pixel (three amounts), and the paged allocator (the k-th needy request
gets the k-th page).  This CPU test states both forms in Python — the reference's loop statement for statement
(rendercache-js/src/Driver.js:252-326, :795-839) and the parallel rule the kernels implement — and compares them on random
states, including frames that request more items than the cache has pages.  (The kernels themselves are pinned to the
reference's frames by tests/test_rc_frames.py and the Node tests; the picks by test_pick_sequence_closed_form_*.)"""
import numpy as np
import pytest


def sequential(w, h, prio, thr, picks_px, pixel_element, item_pixel, item_age, cache_pointer):
    """Driver.directSamples' pick loop body + addRequest + allocCacheItem, in the reference's order"""
    st, n = w + 2, item_pixel.size
    prio, pixel_element, item_pixel, item_age = prio.copy(), pixel_element.copy(), item_pixel.copy(), item_age.copy()
    cleared = np.zeros(n, bool)
    last = h * st + (w if h % 2 == 0 else 1)                    # the pixel the candidate loops leave behind (:813-814)
    requests = []
    for p in picks_px:
        item = pixel_element[p]
        if item < 0:                                            # allocCacheItem (:252-305)
            stop = cache_pointer + 8
            if stop > n:
                stop = 8
            take, oldest, max_age = -1, -1, -1.0
            for i in range(stop - 8, stop):
                if item_pixel[i] < 0:
                    take = i
                    break
                if item_age[i] > max_age:
                    max_age, oldest = item_age[i], i
            if take < 0:
                take = oldest
                cleared[take] = True; item_age[take] = -1
            item_pixel[take] = p; pixel_element[p] = take
            cache_pointer = stop
            item = take
        requests.append(item)
        half = int(np.floor((prio[last] - thr) / 2.0))          # distributeExceedingPriority with the stale pixel (:829-839)
        if half >= 0:
            dx = 1 if (p // st) % 2 == 0 else -1
            prio[p + dx] += half; prio[p + st] += half
    return prio, pixel_element, item_pixel, item_age, cleared, requests, cache_pointer


def parallel(w, h, prio, thr, picks_px, pixel_element, item_pixel, item_age, cache_pointer):
    """what rc_pick / rc_amounts / rc_diffuse / rc_alloc / rc_dirs do, every step a whole-array operation"""
    st, n = w + 2, item_pixel.size
    pages = n // 8
    prio, pixel_element, item_pixel, item_age = prio.copy(), pixel_element.copy(), item_pixel.copy(), item_age.copy()
    cleared = np.zeros(n, bool)
    picks_px = np.asarray(picks_px, dtype=np.int64)
    last = h * st + (w if h % 2 == 0 else 1)
    dx_last = 1 if h % 2 == 0 else -1
    ka = np.flatnonzero(picks_px == last - st); kb = np.flatnonzero(picks_px == last - dx_last)
    ka = int(ka[0]) if ka.size else np.iinfo(np.int32).max
    kb = int(kb[0]) if kb.size else np.iinfo(np.int32).max
    pr = int(prio[last])
    h0 = (pr - thr) >> 1
    if ka != np.iinfo(np.int32).max and h0 >= 0:
        pr += h0
    h1 = (pr - thr) >> 1
    if kb != np.iinfo(np.int32).max and h1 >= 0:
        pr += h1
    halves = np.array([h0, h1, (pr - thr) >> 1])
    k = np.arange(picks_px.size)
    half = halves[np.where(k > kb, 2, np.where(k > ka, 1, 0))]
    on = half >= 0
    dx = np.where((picks_px // st) % 2 == 0, 1, -1)
    np.add.at(prio, (picks_px + dx)[on], half[on]); np.add.at(prio, (picks_px + st)[on], half[on])
    need = pixel_element[picks_px] < 0
    rank = np.cumsum(need) - need
    page0 = (cache_pointer // 8) % pages
    for lo in range(0, picks_px.size, pages):                   # one launch per `pages` needy requests
        sel = np.flatnonzero(need & (rank >= lo) & (rank < lo + pages))
        for kk in sel:                                          # (lanes of one launch: distinct pages, any order)
            base = ((page0 + rank[kk]) % pages) * 8
            slots = np.arange(base, base + 8)
            empty = np.flatnonzero(item_pixel[slots] < 0)
            if empty.size:
                take = slots[empty[0]]
            else:
                take = slots[np.argmax(item_age[slots])]        # first of equals, as the strict > of the reference
                cleared[take] = True; item_age[take] = -1
            item_pixel[take] = picks_px[kk]; pixel_element[picks_px[kk]] = take
    requests = list(pixel_element[picks_px])
    n_need = int(need.sum())
    pointer = ((page0 + n_need) % pages) * 8 if n_need else cache_pointer
    return prio, pixel_element, item_pixel, item_age, cleared, requests, pointer


@pytest.mark.parametrize("w,h,fill,take", [(13, 6, 0.5, 0.3), (8, 5, 0.9, 0.9), (6, 4, 1.0, 1.0), (21, 9, 0.0, 0.6), (5, 2, 0.7, 1.0), (9, 1, 0.5, 1.0)])
def test_parallel_forms_equal_the_sequential_walks(w, h, fill, take):
    st, cells = w + 2, (w + 2) * (h + 2)
    n = w * h * 4 if (w, h) != (6, 4) else 24                   # (6, 4): 3 pages for up to 24 requests: pages are revisited
    pages = n // 8
    for seed in range(40):
        rng = np.random.default_rng(1000 * w + seed)
        prio = np.zeros(cells, np.int64)
        inner = np.zeros((h + 2, w + 2), bool); inner[1:-1, 1:-1] = True
        prio[inner.ravel()] = rng.integers(-30, 256, w * h)
        thr = int(rng.integers(0, 200))
        item_pixel = np.where(rng.random(n) < fill, 0, -1).astype(np.int64)         # attached or empty (the pixel itself is irrelevant here)
        item_age = rng.integers(0, 6, n).astype(np.float64)
        pixel_element = np.full(cells, -1, np.int64)
        owners = rng.permutation(np.flatnonzero(inner.ravel()))[: min(w * h // 2, n)]
        its = rng.permutation(n)[: owners.size]
        pixel_element[owners] = its; item_pixel[its] = owners                       # some pixels already have an item
        # candidates in walk order, a random increasing subset of them picked
        walk = [y * st + x for y in range(1, h + 1) for x in (range(1, w + 1) if y % 2 == 0 else range(w, 0, -1))]
        cand = [p for p in walk if prio[p] >= thr]
        picks = [p for p in cand if rng.random() < take]
        pointer = int(rng.integers(0, pages + 1)) * 8
        a = sequential(w, h, prio, thr, picks, pixel_element, item_pixel, item_age, pointer)
        b = parallel(w, h, prio, thr, picks, pixel_element, item_pixel, item_age, pointer)
        for x, y, name in zip(a[:5], b[:5], ("priority", "pixel_element", "item_pixel", "item_age", "cleared")):
            assert np.array_equal(x, y), (name, seed)
        assert [int(v) for v in a[5]] == [int(v) for v in b[5]], ("requests", seed)
        assert (a[6] // 8) % pages == (b[6] // 8) % pages, ("cache pointer", seed)
