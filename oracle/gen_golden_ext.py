"""TEST INFRASTRUCTURE — regression vectors for the EXTENSION modes (max_depth > 0, spp > 1).

The reference has no such modes, so these files are NOT reference outputs: they
are outputs of the float64 specification in oracle/ptr_oracle.c (orc_render_ext), frozen so that a
later edit of the specification shows up as a diff ("self-pinned, parity unpinned").
Run from the repo root:  python oracle/gen_golden_ext.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import oracle_py as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = [   # scene, width, height, max_depth, spp, seed, keep the frame (else hash only)
    ("example1", 160, 120, 5, 1, 0, True),
    ("example1", 160, 120, 1, 1, 0, True),
    ("example1", 160, 120, 0, 4, 7, True),
    ("example1", 160, 120, 5, 4, 7, True),
    ("example1_permuted", 96, 64, 5, 2, 1, True),
    ("random_3", 96, 64, 5, 1, 0, True),
    ("random_5", 96, 64, 3, 2, 99, True),
    ("inside_sphere", 64, 48, 5, 1, 0, True),
    ("example1", 640, 480, 5, 1, 0, False),
    ("example1", 640, 480, 5, 4, 12345, False),
    ("example1", 1920, 1080, 5, 1, 0, False),
]


# BASELINE.json configs[2] and [3] on the synthetic scene of path-tracer_amd/py/ptr_amd/synthetic.py:
# hashes only (minutes of CPU each); C4 = the rows of ONE part of the 8-way cyclic 8-row strip partition of a 4K frame
CONFIG_SEED = 0x5EED1234
CONFIG_CASES = [   # key, width, height, max_depth, spp, (n_parts, part, strip_height) or None
    ("config2@1920x1080_d5_s64", 1920, 1080, 5, 64, None),
    ("config3_part3of8@3840x2160_d5_s256", 3840, 2160, 5, 256, (8, 3, 8)),
]


def config_cases(index):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "py"))
    from ptr_amd import synthetic as syn          # scene data only (no GPU code is touched)
    soa = syn.config3_scene()
    for key, w, h, depth, spp, strips in CONFIG_CASES:
        if strips is None:
            out = orc.render_ext(soa, syn.BACKGROUND, w, h, max_depth=depth, spp=spp, seed=CONFIG_SEED, threads=8)
        else:
            n_parts, part, sh = strips
            rows = []
            for k in range(part, (h + sh - 1) // sh, n_parts):
                rows.append(orc.render_ext(soa, syn.BACKGROUND, w, h, start_y=k * sh, scan_height=min(sh, h - k * sh),
                                           max_depth=depth, spp=spp, seed=CONFIG_SEED, threads=8))
            out = np.concatenate(rows)
        index["cases"][key] = dict(scene="synthetic.config3_scene", width=w, height=h, max_depth=depth, spp=spp, seed=CONFIG_SEED,
                                   strips=list(strips) if strips else None, pixels=int(out.size),
                                   sha256=hashlib.sha256(out.tobytes()).hexdigest())
        print(key, index["cases"][key]["sha256"][:16], flush=True)


PLANE_CASES = [   # key, width, height, max_depth, spp, seed, keep the frame
    ("planes_phong@160x120_d0_s1", 160, 120, 0, 1, 0, True),
    ("planes_phong@160x120_d3_s2", 160, 120, 3, 2, 21, True),
    ("planes_phong@640x480_d5_s1", 640, 480, 5, 1, 0, False),
]


def plane_cases(index):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "py"))
    from ptr_amd import synthetic as syn          # scene data only
    soa, planes, phong = syn.planes_scene()
    for key, w, h, depth, spp, seed, keep in PLANE_CASES:
        out = orc.render_ext2(soa, syn.BACKGROUND, w, h, planes=planes, phong=phong, max_depth=depth, spp=spp, seed=seed, threads=8)
        entry = dict(scene="synthetic.planes_scene", width=w, height=h, max_depth=depth, spp=spp, seed=seed,
                     sha256=hashlib.sha256(out.tobytes()).hexdigest())
        if keep:
            rel = f"ext/{key}.rgba"
            out.tofile(os.path.join(GOLDEN, rel))
            entry["file"] = rel
        index["cases"][key] = entry
        print(key, entry["sha256"][:16], flush=True)


def main():
    scenes = orc.load_scenes(GOLDEN)
    index = {"note": "outputs of oracle/ptr_oracle.c orc_render_ext (the extension-mode specification); "
                     "not reference outputs", "cases": {}}
    os.makedirs(os.path.join(GOLDEN, "ext"), exist_ok=True)
    for name, w, h, depth, spp, seed, keep in CASES:
        sc = scenes[name]
        soa = orc.scene_to_soa(sc["elements"])
        out = orc.render_ext(soa, sc["background"], w, h, max_depth=depth, spp=spp, seed=seed, threads=8)
        key = f"{name}@{w}x{h}_d{depth}_s{spp}_k{seed}"
        entry = dict(scene=name, width=w, height=h, max_depth=depth, spp=spp, seed=seed,
                     sha256=hashlib.sha256(out.tobytes()).hexdigest())
        if keep:
            rel = f"ext/{key}.rgba"
            out.tofile(os.path.join(GOLDEN, rel))
            entry["file"] = rel
        index["cases"][key] = entry
        print(key, entry["sha256"][:16])
    config_cases(index)
    plane_cases(index)
    with open(os.path.join(GOLDEN, "ext", "index.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
