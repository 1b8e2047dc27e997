"""A host program written in plain C against include/i2r_hip.h (tests/c_abi/abi_smoke.c: HIP runtime for memory, the
C ABI for everything else -- no Python objects, no torch types cross the boundary): delay embedding bit-exact against
loops in C, a 3x3 f16x2 convolution with bias and residual against a double-precision direct convolution, the no-alloc
entry point bit-equal to the allocating one, and the error convention.  Built by __graft_entry__.build(); loaded
in-process here (the GPU boxes refuse an exec from a process that has initialised the GPU)."""
import ctypes
import os

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_c_program_on_the_c_abi():
    so = os.path.join(HERE, "c_abi", "libabi_smoke.so")
    assert os.path.exists(so), "tests/c_abi/libabi_smoke.so is missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(so)
    lib.abi_smoke_run.restype = ctypes.c_int
    assert lib.abi_smoke_run() == 0
