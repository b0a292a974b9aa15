"""The shipped code object must contain no device function call and no unexpected scratch (tools/isa_guard.py: why)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_device_calls_and_bounded_scratch(built):
    import isa_guard
    rep = isa_guard.check()
    print({k: v for k, v in rep.items() if k != "kernels_with_scratch"}, rep["kernels_with_scratch"])
    assert rep["kernels"] > 50
    assert not rep["violations"], rep["violations"]
