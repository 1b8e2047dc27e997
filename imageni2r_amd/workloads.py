"""Named synthetic workloads: the shapes BASELINE.json's configs are quoted on."""
from __future__ import annotations

from types import SimpleNamespace

_COMMON = dict(ema=True, ema_warmup=100, diffusion_steps=18, beta1=1e-5, betaT=1e-2, deterministic=False,
               unet_channels=128)

WORKLOADS = {
    # A: configs/seq_len_24/sine.yaml (CPU-runnable reference case)
    "A_sine24": dict(seq_len=24, delay=3, embedding=8, img_resolution=8, input_channels=5, ch_mult=[1, 2, 2, 2],
                     attn_resolution=[8, 4, 2], batch_size=128, fwd_gflop=1.219),
    # B1: BASELINE.json configs[1] "weather-shaped seq_len=96 (delay 6, embedding 16, img_resolution 8, 12 ch)"
    "B1_weather96": dict(seq_len=96, delay=6, embedding=16, img_resolution=8, input_channels=12,
                         ch_mult=[1, 2, 2, 4], attn_resolution=[16, 8, 4, 2], batch_size=32, fwd_gflop=5.482),
    # C: configs/seq_len_96/electricity.yaml
    "C_electricity96": dict(seq_len=96, delay=6, embedding=16, img_resolution=8, input_channels=321,
                            ch_mult=[1, 2, 2, 4], attn_resolution=[16, 8, 4, 2], batch_size=32, fwd_gflop=5.846),
    # D: configs/seq_len_768/etth1.yaml
    "D_etth1_768": dict(seq_len=768, delay=24, embedding=32, img_resolution=32, input_channels=7,
                        ch_mult=[1, 2, 2, 2], attn_resolution=[32, 16, 8, 4, 2], batch_size=32, fwd_gflop=23.302),
    # The reference's remaining architecture families (not BASELINE shapes; reference-pinned parity cases):
    # F: configs/seq_len_24/mujoco.yaml -- the 64-channel net
    "F_mujoco24": dict(seq_len=24, delay=3, embedding=8, img_resolution=8, input_channels=14, unet_channels=64,
                       ch_mult=[1, 2, 2, 2], attn_resolution=[8, 4, 2], batch_size=128, fwd_gflop=0.311),
    # H: configs/seq_len_768/energy.yaml -- channel multipliers 1, 2, 4, 8 (1024 channels at 4x4)
    "H_energy768": dict(seq_len=768, delay=24, embedding=32, img_resolution=32, input_channels=28,
                        ch_mult=[1, 2, 4, 8], attn_resolution=[32, 16, 8, 4, 2], batch_size=16, fwd_gflop=40.17),
    # I: configs/seq_len_96/weather.yaml -- 96 steps on a 32x32 image (4 used columns), multipliers 1, 2, 2, 4
    "I_weather96_32": dict(seq_len=96, delay=24, embedding=32, img_resolution=32, input_channels=21,
                           ch_mult=[1, 2, 2, 4], attn_resolution=[32, 16, 8, 4, 2], batch_size=32, fwd_gflop=25.33),
}


def workload_args(name: str, device="cuda", **override):
    d = dict(_COMMON)
    d.update(WORKLOADS[name])
    d.update(override)
    d["device"] = device
    return SimpleNamespace(**d)


def image_side(args) -> int:
    """Actual padded image side S = max(embedding, cols), not the nominal img_resolution."""
    from .delay import delay_geometry
    return delay_geometry(args.seq_len, args.delay, args.embedding)[1]
