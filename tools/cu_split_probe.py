#!/usr/bin/env python3
"""Does giving each of the sampler's two half-batch graphs its own set of CUs help?  The two replay streams are made with
hipExtStreamCreateWithCUMask (mask layouts below) instead of plain streams; same graphs, same work.

    python tools/cu_split_probe.py [workload=B1_weather96] [batch=4096] [steps=2] [layouts=none,half,interleave,...]
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageni2r_amd  # noqa: E402
from bench import build_model  # noqa: E402
from imageni2r_amd import DiffusionProcess  # noqa: E402
from imageni2r_amd.workloads import image_side  # noqa: E402

WL = sys.argv[1] if len(sys.argv) > 1 else "B1_weather96"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
LAYOUTS = (sys.argv[4] if len(sys.argv) > 4 else "none,half,interleave,none").split(",")
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for i in bits:
        arr[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device="cuda:0")


def layout(name):
    if name == "none":
        return None
    if name == "half":
        return [range(0, NCU // 2), range(NCU // 2, NCU)]
    if name == "interleave":
        return [range(0, NCU, 2), range(1, NCU, 2)]
    if name == "xcd":  # whole XCDs, if mask bits are dealt round-robin over the 8 XCDs
        return [[i for i in range(NCU) if (i % 8) < 4], [i for i in range(NCU) if (i % 8) >= 4]]
    if name.startswith("overlap"):  # each stream may use 3/4 of the CUs, the middle half is shared
        q = NCU // 4
        return [range(0, 3 * q), range(q, NCU)]
    raise SystemExit(f"unknown layout {name}")


imageni2r_amd.set_conv_math("f16x2")
model, args = build_model(WL, torch.device("cuda"), seed=0)
S = image_side(args)
proc = DiffusionProcess(args, model.net, (args.input_channels, S, S))
xT = torch.randn(B, args.input_channels, S, S, device="cuda")
print("CUs", NCU, "workload", WL, "batch", B, flush=True)
proc.sample(xT)  # capture
for name in LAYOUTS:
    lay = layout(name)
    proc._streams = None if lay is None else [masked_stream(b) for b in lay]
    proc.sample(xT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        proc.sample(xT)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print(f"{name:12s} {B / dt:8.1f} series/s  ({dt * 1e3:.1f} ms per step)", flush=True)
