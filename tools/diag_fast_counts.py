"""Diagnostics (GPU) for the float32 kernel on the example scene at 1080p, from its debug build (PTR_DEBUG_REASONS):
  * shadow tests a wave runs (full tests in bits 8-15, pre-filter tests in bits 16-23 of every pixel);
  * shader clocks a wave spends per stage (debug flag 0x2000: lanes 0..5 of every wave's first row), under load
    (whole frame) and with at most one wave per SIMD (an 8-row band: 120 waves on 1024 SIMDs = pure latency).
Usage: python tools/diag_fast_counts.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np
import ptr_amd

w, h = 1920, 1080
sc = json.load(open(os.path.join(ROOT, "tests/golden/scenes.json")))["example1"]
r = ptr_amd.Renderer(0)
r.set_elements(sc["elements"]); r.set_background(sc["background"])
DBG = ptr_amd.MODE_F32_ONLY | 0x8000


import re
PPL = int(re.search(r"define PTR_FAST_PPL (\d+)", open(os.path.join(ROOT, "path-tracer_amd/csrc/ptr_device.h")).read()).group(1))
TW = 8 * PPL


def tiles(img, rows):
    """one row of values per wave tile (8 * PPL wide x 8 tall, the pixel slots side by side): the tile's first pixel row"""
    return img[: rows // 8 * 8].reshape(rows // 8, 8, w // TW, TW)[:, 0]


why = r.render(w, h, flags=DBG).reshape(h, w)
tf = tiles((why >> 8) & 0xFF, h)[..., 0]
tp = tiles((why >> 16) & 0xFF, h)[..., 0]
sel = tp > 0
print("waves", tf.size, " waves that ran shadow tests", int(sel.sum()))
print("pre-filter tests per such wave: mean %.2f" % tp[sel].mean(), " full tests: mean %.2f" % tf[sel].mean())
print("histogram of full tests per such wave:", np.bincount(tf[sel].ravel(), minlength=8)[:10])
print("flagged pixels", int(((why & 0xFF) != 0).sum()))

names = ["set-up", "cull", "nearest hit", "winner", "lights", "pack"]


def stages(img, rows, label):
    tt = tiles(img, rows)[..., :6].astype(np.int64)
    ok = (tt[..., 4] < 1_000_000) & (tt[..., 5] < 1_000_000) & (tt[..., 4] > 0)      # waves that shaded carry all stamps
    print(label, int(ok.sum()), "shading waves, clocks per stage:", {n_: int(tt[..., k][ok].mean()) for k, n_ in enumerate(names)},
          " total", int(tt[ok].sum(axis=-1).mean()))


stages(r.render(w, h, flags=DBG | 0x2000).reshape(h, w), h, "loaded (whole frame):")
for y0 in (300, 700):
    stages(r.render(w, h, start_y=y0, scan_height=8, flags=DBG | 0x2000).reshape(8, w), 8,
           "one wave per SIMD at most, rows %d..%d:" % (y0, y0 + 7))
