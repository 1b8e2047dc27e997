"""The driver-timed workloads AT THE BENCH BATCH: B1 (weather-96) at 4096 series per step and D (ETTh1-768) at 1024.

Only at these sizes does the default dispatch select, all at once, the persistent 1x1 kernel (>= 65536 positions), the
256-position 3x3 tiles, the many-small-groups GroupNorm, the chunked attention tail and the two concurrent half-batch
graphs.  Each has a unit test; this file checks their COMPOSITION against what the reference pins:

* the sampler at the bench batch (`DiffusionProcess.sample`, models/sampler.py:32-71) gives, for the golden latent put
  at both ends of the batch, the reference's own sampled series (tests/golden/g7_fullsize.npz, 1e-4), and its first
  two series equal the same two latents sampled as a batch of 2 -- the golden-pinned small-batch path -- within 1e-5;
* one denoiser evaluation (models/networks.py:654-668) at the bench batch equals the same samples evaluated in
  chunks of 16 (1e-5), for a 0-dim sigma (the sampler's form) and for per-sample sigmas."""
import numpy as np
import pytest
import torch

from oracle import i2r_oracle as O
from tests.helpers import assert_close, fullsize_inputs, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", params=[("B1_weather96", 4096), ("D_etth1_768", 1024)], ids=["B1_b4096", "D_b1024"])
def big(request):
    import imageni2r_amd
    from imageni2r_amd import TS2img_Karras
    name, batch = request.param
    args, S, d = fullsize_inputs(name)
    args.device = DEV
    torch.manual_seed(0)
    m = TS2img_Karras(args, DEV)
    sd = m.net.state_dict()
    O.perturb_zero_init(sd, seed=1234)     # the weights of the g7 goldens
    O.perturb_norm_affine(sd, seed=4321)
    imageni2r_amd.set_conv_math("f16x2")
    yield name, batch, m.to(DEV).eval(), args, S, d
    del m
    torch.cuda.empty_cache()


def test_sampler_at_the_bench_batch_against_the_golden_and_the_small_batch_path(big):
    from imageni2r_amd import DiffusionProcess
    name, batch, m, args, S, d = big
    g = load_golden("g7_fullsize.npz")
    C = args.input_channels
    gen = torch.Generator().manual_seed(77)
    xT = torch.randn(batch, C, S, S, generator=gen)
    xT[0] = d["xT"][0]
    xT[batch - 1] = d["xT"][0]
    proc = DiffusionProcess(args, m.net, (C, S, S))       # default dispatch, as bench.py's sampler_bench builds it:
    assert proc._dual(batch)                              # B1 at 4096 and D at 1024 (a million positions each): two
                                                          # concurrent half-batch graphs
    assert not proc._dual(batch // 4) and not proc._dual(batch - 1)
    out = proc.sample(xT)
    assert out.dtype == torch.float64 and tuple(out.shape) == (batch, C, S, S) and bool(torch.isfinite(out).all())
    ref = g[name + "/sample"]
    assert rel_err(out[:1].cpu().numpy(), ref) < 1e-4, "first series of the first half-batch graph"
    assert rel_err(out[batch - 1:].cpu().numpy(), ref) < 1e-4, "last series of the second half-batch graph"
    for _ in range(2):   # (two graphs side by side: how the store-data hazard of the small attention kernel showed)
        assert torch.equal(out, proc.sample(xT)), "a replay repeats to the bit"
    small = DiffusionProcess(args, m.net, (C, S, S)).sample(xT[:2])   # the batch the goldens pin
    assert_close(out[:2].cpu().numpy(), small.cpu().numpy(), 1e-5, "series [0:2]: bench batch vs batch 2")
    tail = DiffusionProcess(args, m.net, (C, S, S)).sample(xT[batch - 2:])
    assert_close(out[batch - 2:].cpu().numpy(), tail.cpu().numpy(), 1e-5, "series [-2:]: bench batch vs batch 2")
    # the inverse transform of the whole batch: every series' bits equal the oracle's on the same image
    m.ts_to_img(torch.rand(batch, args.seq_len, C).to(DEV))
    ts = m.img_to_ts(out)
    pick = [0, 1, batch // 2, batch - 1]
    assert np.array_equal(ts[pick].cpu().numpy(),
                          O.img_to_ts(out[pick].cpu().numpy(), args.seq_len, args.delay, args.embedding))


def test_denoiser_at_the_bench_batch_equals_chunks_of_16(big):
    name, batch, m, args, S, d = big
    C = args.input_channels
    gen = torch.Generator().manual_seed(78)
    x = (torch.randn(batch, C, S, S, generator=gen) * 1.3).to(DEV)
    sig0 = torch.tensor(1.7, dtype=torch.float64)
    sigB = torch.exp(torch.randn(batch, generator=gen) * 1.2 - 1.2).to(DEV)
    with torch.no_grad():
        for sig, what in ((sig0, "0-dim sigma"), (sigB, "per-sample sigma")):
            whole = m.net(x, sig)
            parts = torch.cat([m.net(x[i:i + 16], sig if sig.ndim == 0 else sig[i:i + 16])
                               for i in range(0, batch, 16)])
            assert_close(whole.cpu().numpy(), parts.cpu().numpy(), 1e-5, (name, batch, what))
            del whole, parts
