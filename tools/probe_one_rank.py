"""bench.py's exchange probe (the child process of a rank at N > 1) with ONE rank, where the library needs no RCCL: does the
probe's own code run on this box?   python tools/probe_one_rank.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import bench  # noqa: E402
bench.exchange_probe_child(bytes(128).hex(), 0, 1, 0)
