"""The JavaScript CPU port (oracle/js: bench.py's "reference-style JS under worker_threads" baseline) is
pinned to the golden frames of the unmodified reference, and its worker_threads driver runs."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def test_js_port_reproduces_the_reference_frames():
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "cpu_check.mjs")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0, out.stdout.decode()
    assert out.stdout.decode().startswith("ok ")


def test_worker_threads_driver_counts_the_reference_ray_mix():
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "cpu_bench.mjs"), "2", "0.2", "640", "480"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    r = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert r["threads"] == 2 and r["frames"] >= 1
    import oracle_py as orc
    with open(os.path.join(ROOT, "tests", "golden", "scenes.json")) as f:
        sc = json.load(f)["example1"]
    _, cnt = orc.render(orc.scene_to_soa(sc["elements"]), sc["background"], 640, 480, flags=0, counters=True)
    assert r["rays"] == r["frames"] * (cnt["primary"] + cnt["shadow_rays"])     # same ray mix as the C oracle
