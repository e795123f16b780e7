"""Diagnostic (debug build -DPTR_EXT_CHECK_DEBUG): which criterion of the checked bounce loop flags the pixels of configs[1]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np, ptr_amd
from bench import example1_soa, WIDTH, HEIGHT
r = ptr_amd.Renderer(0)
r.set_scene(example1_soa()); r.set_background([2.0, 2.0, 2.0])
for depth in (1, 2, 3, 5):
    why = r.render(WIDTH, HEIGHT, flags=ptr_amd.MODE_TOLERANT, max_depth=depth) & 0xffff
    print("depth", depth, "flagged %.2f %%" % (100.0 * np.count_nonzero(why) / why.size),
          {b: round(100.0 * int(((why >> b) & 1).sum()) / why.size, 2) for b in range(8)})
