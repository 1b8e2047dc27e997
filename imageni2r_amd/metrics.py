"""Evaluation metrics of the run's evaluation stage, kept ON THE DEVICE.

The reference scores every evaluation epoch with a post-hoc GRU discriminator (metrics/discriminative_torch.py:57-133),
and -- when that improves -- a GRU one-step predictor (metrics/predictive_metrics.py:116-220) and a cross-correlation
distance (metrics/correlation_score.py:117-136), dispatched by metrics/__init__.py:8-52.  These gate checkpointing
(run_irregular.py:267-284) and are the acceptance measure BASELINE.json attaches to the reduced-precision path.
They are plain PyTorch and stay so (torch.nn.GRU on PyTorch-ROCm: MIOpen's RNN) -- no new kernels; what changes is the
step boundary: the series stay on the device they were generated on instead of being re-stacked from Python lists
and copied host->device twice per iteration (2 x 2000 copies), the per-iteration `d_loss.cpu().item()` sync
(:112, its sum is never used) is gone, and the predictor's test pass is one batched call instead of one call per series.

Same algorithms, same hyper-parameters, same random-number consumption order (numpy's global generator for the
splits / mini-batches, torch's for the network initialisation): with the same seeds on the same device the
discriminative score is bit-identical to the reference's (tests/test_metrics_cpu.py checks that in the build
container against the imported reference).  Context-FID (metrics/context_fid.py: a TS2Vec encoder fit) lives in
imageni2r_amd/context_fid.py and is dispatched from here.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _as_device_tensor(data, device):
    if isinstance(data, torch.Tensor):
        return data.to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(data), dtype=torch.float32).to(device)


def _graph_loop_enabled(x):
    return x.is_cuda and os.environ.get("I2R_METRICS_GRAPH", "1") != "0"


def _train_graphed(model, loss_of, set_sizes, iterations, batch_size, device):
    """The 2000-step mini-batch loop of a scoring network as ONE captured HIP graph replayed per iteration: the loop
    is launch-latency bound (a GRU with a handful of hidden units), so capture is what a GPU can do for it.  Same
    algorithm and the same numpy draws in the same order -- every iteration's two index permutations are drawn up
    front and copied to the device once; Adam runs in torch's `capturable` form (its step counter on the device).
    The three eager warm-up steps torch asks for before a capture are undone (weights restored, moments and step
    counter zeroed in place).  Returns False -- with the model untouched -- when the step cannot be captured."""
    params = [p for p in model.parameters()]
    saved = [p.detach().clone() for p in params]
    sizes = [min(batch_size, n) for n in set_sizes]
    idx = [torch.zeros(k, dtype=torch.long, device=device) for k in sizes]
    optimizer = torch.optim.Adam(params, capturable=True)
    graph = torch.cuda.CUDAGraph()
    try:
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                optimizer.zero_grad(set_to_none=True)
                loss_of(*idx).backward()
                optimizer.step()
        torch.cuda.current_stream(device).wait_stream(side)
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss_of(*idx).backward()
            optimizer.step()
    except Exception as e:  # e.g. a library call that cannot be captured: train eagerly instead
        warnings.warn(f"metrics: graph capture of the training step failed ({type(e).__name__}: {e}); eager loop")
        with torch.no_grad():
            for p, s0 in zip(params, saved):
                p.copy_(s0)
        return False
    with torch.no_grad():
        for p, s0 in zip(params, saved):
            p.copy_(s0)
        for st in optimizer.state.values():
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["step"].zero_()
    draws = np.stack([np.concatenate([np.random.permutation(n)[:k] for n, k in zip(set_sizes, sizes)])
                      for _ in range(iterations)]) if iterations else np.zeros((0, sum(sizes)), dtype=np.int64)
    all_idx = torch.as_tensor(draws, dtype=torch.long).to(device)
    for i in range(iterations):
        off = 0
        for buf, k in zip(idx, sizes):
            buf.copy_(all_idx[i, off:off + k])
            off += k
        graph.replay()
    torch.cuda.current_stream(device).synchronize()  # the graph's buffers die with this frame
    return True


class Discriminator(nn.Module):
    """1-layer GRU over (B, L, C) + Linear on the last hidden state (discriminative_torch.py:67-81)."""

    def __init__(self, inp_dim, hidden_dim):
        super().__init__()
        self.rnn = nn.GRU(input_size=inp_dim, hidden_size=hidden_dim, bidirectional=False, num_layers=1,
                          batch_first=True)
        self.linear = nn.Linear(hidden_dim, 1)

    def forward(self, x):
        _, last_hidden_state = self.rnn(x)
        y_hat_logit = self.linear(last_hidden_state)
        return y_hat_logit, torch.sigmoid(y_hat_logit)


def discriminative_score_metrics(ori_data, generated_data, args, iterations=2000, batch_size=32):
    """|0.5 - accuracy| of a post-hoc real-vs-generated classifier (discriminative_torch.py:57-133).

    Random draws, in the reference's order: torch (network init), then numpy: the two 80/20 split permutations
    (:18, :27), then per iteration one permutation for the real and one for the generated mini-batch (:49)."""
    device = args.device
    ori = _as_device_tensor(ori_data, device)
    gen = _as_device_tensor(generated_data, device)
    hidden_dim = 8 if args.input_size == 1 else int(args.input_size / 2)
    model = Discriminator(args.input_size, hidden_dim).to(device)

    def split(n, train_rate=0.8):
        idx = np.random.permutation(n)
        return idx[:int(n * train_rate)], idx[int(n * train_rate):]

    tr_o, te_o = split(len(ori))
    tr_g, te_g = split(len(gen))
    train_x, train_x_hat = ori[torch.as_tensor(tr_o, device=device)], gen[torch.as_tensor(tr_g, device=device)]
    test_x, test_x_hat = ori[torch.as_tensor(te_o, device=device)], gen[torch.as_tensor(te_g, device=device)]

    model.train()

    def d_loss_of(ia, ib):
        y_logit_real, _ = model(train_x[ia])
        y_logit_fake, _ = model(train_x_hat[ib])
        return (F.binary_cross_entropy_with_logits(y_logit_real, torch.ones_like(y_logit_real)).mean()
                + F.binary_cross_entropy_with_logits(y_logit_fake, torch.zeros_like(y_logit_fake)).mean())

    if not (_graph_loop_enabled(train_x) and
            _train_graphed(model, d_loss_of, (len(train_x), len(train_x_hat)), iterations, batch_size, train_x.device)):
        optimizer = torch.optim.Adam(model.parameters())
        for _ in range(iterations):
            ia = torch.as_tensor(np.random.permutation(len(train_x))[:batch_size], device=device)
            ib = torch.as_tensor(np.random.permutation(len(train_x_hat))[:batch_size], device=device)
            d_loss = d_loss_of(ia, ib)
            optimizer.zero_grad()
            d_loss.backward()
            optimizer.step()

    model.eval()
    with torch.no_grad():
        _, p_real = model(test_x)
        _, p_fake = model(test_x_hat)
        correct = (p_real.reshape(-1) > 0.5).sum() + (p_fake.reshape(-1) <= 0.5).sum()
        acc = correct.item() / (p_real.numel() + p_fake.numel())
    return np.abs(0.5 - acc)


class GRUPredictorIrregular(nn.Module):
    """GRU(dim -> dim/2) + Linear(dim/2 -> dim) + sigmoid (predictive_metrics.py:45-86); every series here has the
    full length, so the reference's pack / unpack of the sequences is the identity."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.gru = nn.GRU(input_size=input_dim, hidden_size=hidden_dim, batch_first=True)
        self.dense = nn.Linear(hidden_dim, input_dim)

    def forward(self, X):
        out, _ = self.gru(X)
        return torch.sigmoid(self.dense(out))


def predictive_score_metrics(ori_data, generated_data, device="cpu", iterations=2, test_batch=4096):
    """MAE of a one-step-ahead predictor trained on the generated series and tested on the original ones
    (predictive_metrics.py:116-220; the reference trains for 2 iterations and keeps it all on the CPU)."""
    ori = _as_device_tensor(ori_data, device)
    gen = _as_device_tensor(generated_data, device)
    no, seq_len, dim = ori.shape
    hidden_dim = int(dim / 2)
    if seq_len > 96:
        batch_size = 4 if dim > 10 else 16
    else:
        batch_size = 128
    model = GRUPredictorIrregular(dim, hidden_dim).to(device)
    solver = torch.optim.Adam(model.parameters())
    for _ in range(iterations):
        idx = torch.as_tensor(np.random.permutation(len(gen))[:batch_size], device=device)
        mb = gen[idx]
        loss = F.l1_loss(model(mb[:, :-1]), mb[:, 1:])
        solver.zero_grad()
        loss.backward()
        solver.step()
    np.random.permutation(no)  # the reference draws (and then does not need) a test order (:189); keeps streams aligned
    total = torch.zeros((), dtype=torch.float64, device=device)
    with torch.no_grad():
        for i in range(0, no, test_batch):
            x = ori[i:i + test_batch]
            total += (model(x[:, :-1]) - x[:, 1:]).abs().double().mean(dim=(1, 2)).sum()  # per-series MAE, summed
    return float(total.item() / no)


def cross_correlation_features(x):
    """Lag-0 cross-correlation of every feature pair (lower triangle incl. diagonal), features standardised over
    (series, time) with the unbiased std -- cacf_torch(x, max_lag=1).mean(0)[0] of correlation_score.py:18-87 as
    ONE matrix product instead of a Python loop over index chunks."""
    B, T, C = x.shape
    z = (x - x.mean(dim=(0, 1), keepdim=True)) / x.std(dim=(0, 1), keepdim=True)
    z = z.reshape(B * T, C)
    corr = (z.t() @ z) / (B * T)
    i, j = torch.tril_indices(C, C, device=x.device)
    return corr[i, j]


def calculate_pearson_correlation(real_sig, gen_sig, device="cpu"):
    """sum |corr_fake - corr_real| / 10 over the feature pairs, on equally sized random resamples (with replacement)
    of both sets (correlation_score.py:117-136); returns (mean, std, conf) like display_scores with one iteration."""
    real = _as_device_tensor(real_sig, device).double()
    fake = _as_device_tensor(gen_sig, device).double()
    size = int(real.shape[0])
    real_idx = torch.as_tensor(np.random.randint(low=0, high=real.shape[0], size=(size,)), device=device)
    fake_idx = torch.as_tensor(np.random.randint(low=0, high=fake.shape[0], size=(size,)), device=device)
    diff = cross_correlation_features(fake[fake_idx]) - cross_correlation_features(real[real_idx])
    score = float((diff.abs().sum() / 10.0).item())
    return round(score, 4), float("nan"), float("nan")  # one iteration: std / confidence interval are undefined


def evaluate_model_irregular(real_sig, gen_sig, args, calc_other_metrics=False, metric_iteration=1):
    """Dispatcher with the reference's keys (metrics/__init__.py:8-52).  ``calc_other_metrics``: predictive score,
    context-FID and correlation score, evaluated in the reference's order (they share the random generators)."""
    if calc_other_metrics:
        from .context_fid import calculate_fid
        pred = [predictive_score_metrics(real_sig, gen_sig, device=args.device) for _ in range(metric_iteration)]
        pred_mean, pred_std = np.round(np.mean(pred), 4), np.round(np.std(pred), 4)
        fid_mean, fid_std, fid_ci = calculate_fid(real_sig, gen_sig, device=args.device)
        corr_mean, corr_std, corr_ci = calculate_pearson_correlation(real_sig, gen_sig, device=args.device)
        return {'pred_score_mean': pred_mean, 'pred_score_std': pred_std, 'fid_score_mean': fid_mean,
                'fid_score_std': fid_std, 'fid_score_conf_interval': fid_ci, 'correlation_score_mean': corr_mean,
                'correlation_score_std': corr_std, 'correlation_score_conf_interval': corr_ci}
    disc = [discriminative_score_metrics(real_sig, gen_sig, args) for _ in range(metric_iteration)]
    return {'disc_mean': np.round(np.mean(disc), 4), 'disc_std': np.round(np.std(disc), 4)}
