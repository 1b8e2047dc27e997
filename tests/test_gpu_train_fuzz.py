"""90 random problems through the TRAINING kernels (tools/train_fuzz.py: conv forward / data gradient in fp32 and f16x2
form / weight + bias gradient in atomic and deterministic form, GroupNorm forward / backward in its fused modes,
attention forward / backward) against torch autograd in fp64 at 3e-5 (max-relative and RMS-relative)."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_training_kernel_problems(monkeypatch):
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "train_fuzz.py")
    monkeypatch.setattr(sys, "argv", [tool, "90", "21"])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(tool, run_name="__main__")
    assert e.value.code == 0
