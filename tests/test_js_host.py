"""The JavaScript host layer (path-tracer_amd/js): the reference's class API over the N-API addon.
CPU part: classes, wire format, flatten order, no-CPU-fallback.  GPU part: renders through the
addon and compares with the golden vectors of the unmodified reference."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "path-tracer_amd", "js")
NODE = shutil.which("node")


def _run(mode):
    assert os.path.exists(os.path.join(JS, "addon", "ptr_napi.node")), \
        "addon not built: make -C path-tracer_amd/js/addon (or __graft_entry__.build())"
    r = subprocess.run([NODE, os.path.join(JS, "test", "run_tests.mjs"), mode], capture_output=True,
                       text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_js_host_logic_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    out = _run("cpu")
    assert "8 JS tests passed" in out


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_js_drop_in_gpu():
    out = _run("gpu")
    assert "JS tests passed (gpu)" in out
