"""Diagnostics (GPU) for the many-sphere kernel: reasons for re-tracing and level-2 call counts.
Usage: python tools/diag_many.py [w] [h] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np
import ptr_amd
from ptr_amd import synthetic as syn

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
r = ptr_amd.Renderer(0)
r.set_scene(syn.many_spheres(n)); r.set_background(syn.BACKGROUND)
out = r.render(w, h, flags=ptr_amd.MODE_F32_ONLY | ptr_amd.DEBUG_REASONS | ptr_amd.DEBUG_FORCE_MANY)
why = out & 0xff
names = {1: "primary band", 2: "origin on surface", 4: "ordering", 8: "far clip", 16: "shadow band", 32: "rounding band"}
print(f"{n} spheres {w}x{h}: flagged {np.count_nonzero(why)} of {w*h} ({100*np.count_nonzero(why)/(w*h):.3f}%)")
for b, nm in names.items():
    print(f"  {nm:18s} {int(((why & b) != 0).sum()):8d}   only-this {int((why == b).sum()):8d}")
l2p = ((out >> 8) & 0xfff).astype(np.float64); l2s = ((out >> 20) & 0xfff).astype(np.float64)
print(f"level-2 calls per wave (256 pixels): primary mean {l2p.mean():.1f} max {l2p.max():.0f};  shadow mean {l2s.mean():.1f} max {l2s.max():.0f}")
