"""The float64 bounce loop's claim "what JavaScript's Math.sqrt and `/` return" rests on the GPU's float64 square root and
division being correctly rounded: the compiler's own sequences, and sqrt_mid / recip_mid of ptr_f64_roots.h (the same sequences
without the range patches, used behind a wave-level exponent check).  tools/ubench/f64_roots (built by __graft_entry__.build()
from that very header) evaluates them on the GPU on the classic hard cases - significands of all ones at 801 exponents, powers
of two and their neighbours, 1 +- 4 096 ulps, 4 M squares and near-tie squares - and on random operands, and compares every
result with the host's (x86 SSE2: IEEE).  Bit-exact is the bar."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "ubench", "f64_roots")


@pytest.mark.gpu
def test_float64_roots_and_reciprocals_are_the_ieee_ones():
    if not os.path.exists(EXE):
        pytest.skip("tools/ubench/f64_roots is not built (python __graft_entry__.py builds it)")
    out = subprocess.run([EXE, "16"], capture_output=True, text=True, timeout=110)
    tail = "\n".join(out.stdout.splitlines()[-24:])
    assert out.returncode == 0, tail + out.stderr[-400:]
    line = [l for l in out.stdout.splitlines() if l.startswith("total differing results")]
    assert line and line[0].rstrip().endswith(": 0"), tail
    # the hard cases were really there, and the dropped candidate still shows why it was dropped (the check can tell right from wrong)
    hard = [l for l in out.stdout.splitlines() if l.startswith("hard cases")]
    assert hard and " 205056 operands: 0 0 0 0 0 800" in hard[0], hard
