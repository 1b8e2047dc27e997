import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.helpers import tiny_args  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tiny_state():
    """Tiny net built by OUR constructor on CPU with the golden seed, zero-init layers perturbed like the
    generator did (tests/golden/make_golden.py:build_tiny).  Valid because test_host_logic pins the init order."""
    import torch
    from imageni2r_amd import TS2img_Karras
    from oracle.i2r_oracle import perturb_zero_init, unet_spec
    torch.manual_seed(0)
    m = TS2img_Karras(tiny_args(), "cpu")
    perturb_zero_init(m.net.state_dict(), seed=1234)
    a = tiny_args()
    spec = unet_spec(a.img_resolution, a.input_channels, a.unet_channels, a.ch_mult, a.attn_resolution)
    return m, spec
