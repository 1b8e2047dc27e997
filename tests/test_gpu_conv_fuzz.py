"""150 random conv problems (tools/conv_fuzz.py: kernel size, shapes down to 1x1 images and 3 channels, pair-format
input, residual, upsampled input / residual, qkv output, bias or none -- every dispatch branch of i2r_conv2d_fwd_h2)
against torch in fp64 at 2e-5, each output inside a guard band."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_conv_problems(monkeypatch):
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "conv_fuzz.py")
    monkeypatch.setattr(sys, "argv", [tool, "150", "11"])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(tool, run_name="__main__")
    assert e.value.code == 0
