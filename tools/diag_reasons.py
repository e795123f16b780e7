"""Diagnostics (GPU): why does the hybrid mode re-trace pixels, and how far is the float32
colour from the float64 oracle?  Usage: python tools/diag_reasons.py [scene] [w] [h]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "path-tracer_amd", "py")]
import numpy as np
import oracle_py as orc
import ptr_amd

name = sys.argv[1] if len(sys.argv) > 1 else "example1"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 640
h = int(sys.argv[3]) if len(sys.argv) > 3 else 480
sc = json.load(open(os.path.join(ROOT, "tests/golden/scenes.json")))[name]
r = ptr_amd.Renderer(0)
r.set_elements(sc["elements"]); r.set_background(sc["background"])
why = r.render(w, h, flags=ptr_amd.MODE_F32_ONLY | 0x8000)
names = {1: "primary band", 2: "origin on surface", 4: "ordering", 8: "far clip", 16: "shadow band", 32: "rounding band"}
print(f"{name} {w}x{h}: flagged {np.count_nonzero(why)} of {w*h} ({100*np.count_nonzero(why)/(w*h):.3f}%)")
for b, nm in names.items():
    print(f"  {nm:18s} {int(((why & b) != 0).sum()):8d}   only-this {int((why == b).sum()):8d}")
wm = why.reshape(h, w)
for k in range(8):
    band = wm[k * h // 8:(k + 1) * h // 8]
    print(f"  rows {k * h // 8:4d}-{(k + 1) * h // 8 - 1:4d}: rounding-only {int((band == 32).sum()):7d}  any {int((band != 0).sum()):7d}  of {band.size}")
soa = orc.scene_to_soa(sc["elements"])
want = orc.render_rgb64(soa, sc["background"], w, h)
got8, f32 = r.render(w, h, flags=ptr_amd.MODE_F32_ONLY, want_f32=True)
want8 = orc.render(soa, sc["background"], w, h, flags=0, threads=8)
err = np.abs(f32[..., :3].astype(np.float64) - want).max(axis=2)
flag = (why != 0).reshape(h, w)
print("colour |err|: max over unflagged", err[~flag].max(), " p99.9", np.quantile(err[~flag], 0.999),
      " max over all", err.max())
print("unflagged pixels whose bytes differ:", int(((got8 != want8).reshape(h, w) & ~flag).sum()),
      " flagged pixels whose bytes differ:", int(((got8 != want8).reshape(h, w) & flag).sum()))
# distance of v*255 from the rounding boundary among unflagged pixels vs the actual error
v = np.minimum(want, 1.0) * 255.0
print("max colour error in LSB units (unflagged):", (np.abs(np.minimum(f32[..., :3].astype(np.float64), 1) * 255 - v).max(axis=2))[~flag].max())
diff = (got8 != want8).reshape(h, w)
whym = why.reshape(h, w)
for b, nm in names.items():
    sel = (whym & b) != 0
    print(f"  diff among {nm:18s}: {int((diff & sel).sum())}")
# which element do shadow-flagged pixels hit?  (oracle, per pixel)
ys, xs = np.nonzero((whym & 16) != 0)
angle = orc.lib().orc_reference_angle()
from collections import Counter
cnt = Counter()
for y, x in list(zip(ys, xs))[:4000]:
    xx = (2 * ((x + 0.5) / w) - 1) * angle * (w / h); yy = (1 - 2 * ((y + 0.5) / h)) * angle
    d = np.array([xx, yy, -1.0]); d /= np.sqrt((d * d).sum())
    _, hit = orc.trace(soa, sc["background"], [0, 0, 0], d, want_hit=True)
    cnt[(hit["element"], bool(diff[y, x]))] += 1
print("shadow-flagged pixels by (hit element, differs):", sorted(cnt.items()))
sel = diff & ((whym & 16) != 0)
ys, xs = np.nonzero(sel)
if ys.size: print("differing shadow-flagged pixels: rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
g = got8.reshape(h, w); wv = want8.reshape(h, w)
for y, x in list(zip(ys, xs))[:40]:
    print(y, x, hex(g[y, x]), hex(wv[y, x]), f32[y, x, :3], want[y, x])
hy = r.render(w, h, flags=0)
bad = np.flatnonzero(hy != want8)
print("HYBRID differs in", bad.size, "pixels; of these flagged:", int(flag.ravel()[bad].sum()), " retraced stat:", r.stats()["retraced"])
for i in bad[:10]:
    yy_, xx_ = divmod(int(i), w)
    print("  ", yy_, xx_, hex(hy[i]), hex(want8[i]), hex(got8[i]), "why", int(why[i]))
