"""Context-FID: the Frechet distance between TS2Vec representations of the real and the generated series
(reference: metrics/context_fid.py:655-684 `Context_FID` / `calculate_fid`, which fit the
TS2Vec encoder vendored in the same file, :143-637, on the real set and embed both sets with it).

Plain PyTorch, like the other evaluation scores (imageni2r_amd/metrics.py): it runs wherever `device` says, which for
the evaluation stage is the GPU the series were generated on (the reference hard-codes `device=0`).  Only what
`Context_FID` reaches is built: the contrastive fit with its default schedule (200 iterations for <= 100000 input
numbers, else 600; batch 8; AdamW 1e-3; random overlapping crops; binomial timestamp masking; hierarchical
instance + temporal contrast over max-pooled scales; a running average of the weights, which is what encodes), and the
`full_series` encoding (max over time).  Sliding-window / multiscale / causal encoding and save/load are not.

Same algorithm, same hyper-parameters, and the same random-number consumption order as the reference -- torch's
global generator: layer initialisation in construction order, the loader's shuffles, dropout; numpy's: the crop
geometry (five draws per iteration), two binomial masks per iteration, the final row permutation -- so that with equal
seeds on the same device type the score reproduces the reference's (tests/test_context_fid_cpu.py, fixture
tests/golden/g9_context_fid.json made from the imported reference on the CPU).
"""
from __future__ import annotations

import warnings

import numpy as np
import scipy.linalg
import scipy.stats
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.data import DataLoader, TensorDataset


class _LengthKeepingConv(nn.Module):
    """Dilated Conv1d whose output has the input's length (context_fid.py:143-160): pad by half the receptive field
    on both sides and drop the surplus last step when the receptive field is even."""

    def __init__(self, cin, cout, kernel, dilation):
        super().__init__()
        field = (kernel - 1) * dilation + 1
        self.conv = nn.Conv1d(cin, cout, kernel, padding=field // 2, dilation=dilation)
        self.surplus = 1 - field % 2

    def forward(self, x):
        y = self.conv(x)
        return y[:, :, :-self.surplus] if self.surplus else y


class _ResidualBlock(nn.Module):
    """gelu -> conv -> gelu -> conv, plus the (1x1-projected where the width changes, and always in the last block)
    input (context_fid.py:163-176).  Construction order conv1, conv2, projector fixes the initialisation draws."""

    def __init__(self, cin, cout, kernel, dilation, last):
        super().__init__()
        self.conv1 = _LengthKeepingConv(cin, cout, kernel, dilation)
        self.conv2 = _LengthKeepingConv(cout, cout, kernel, dilation)
        self.projector = nn.Conv1d(cin, cout, 1) if (cin != cout or last) else None

    def forward(self, x):
        skip = x if self.projector is None else self.projector(x)
        return self.conv2(F.gelu(self.conv1(F.gelu(x)))) + skip


class SeriesEncoder(nn.Module):
    """(B, T, C) -> (B, T, output_dims): Linear per timestamp, timestamp masking, `depth` + 1 residual blocks of
    dilation 1, 2, 4, ... and dropout 0.1 on the representations (context_fid.py:222-269)."""

    def __init__(self, input_dims, output_dims, hidden_dims=64, depth=10):
        super().__init__()
        self.input_fc = nn.Linear(input_dims, hidden_dims)
        widths = [hidden_dims] * depth + [output_dims]
        self.blocks = nn.Sequential(*[
            _ResidualBlock(hidden_dims if i == 0 else widths[i - 1], widths[i], 3, 2 ** i, i == len(widths) - 1)
            for i in range(len(widths))])
        self.repr_dropout = nn.Dropout(p=0.1)

    def forward(self, x, mask=None):
        """mask: None = 'binomial' (p = 0.5 per timestamp, numpy's generator) in training mode and all timestamps in
        eval mode; or the string 'binomial' / 'all_true'.  Timestamps with any NaN feature are zeroed and masked."""
        observed = ~x.isnan().any(dim=-1)
        x = self.input_fc(torch.where(observed.unsqueeze(-1), x, torch.zeros((), dtype=x.dtype, device=x.device)))
        if mask is None:
            mask = "binomial" if self.training else "all_true"
        if mask == "binomial":
            keep = torch.from_numpy(np.random.binomial(1, 0.5, size=(x.size(0), x.size(1)))).to(torch.bool)
            keep = keep.to(x.device) & observed
        elif mask == "all_true":
            keep = observed
        else:
            raise ValueError(f"SeriesEncoder: mask mode {mask!r} is not built (binomial, all_true)")
        x = torch.where(keep.unsqueeze(-1), x, torch.zeros((), dtype=x.dtype, device=x.device))
        return self.repr_dropout(self.blocks(x.transpose(1, 2))).transpose(1, 2)


def _off_diagonal_nll(sim):
    """-log_softmax of every row of a batch of square similarity matrices over its OFF-diagonal entries
    (.., n, n) -> (.., n, n - 1): column j of the result is entry j below the diagonal and entry j + 1 above it."""
    logits = torch.tril(sim, diagonal=-1)[..., :-1] + torch.triu(sim, diagonal=1)[..., 1:]
    return -F.log_softmax(logits, dim=-1)


def instance_contrast(z1, z2):
    """Per timestamp: the two views of a series are the positive pair among the 2B views of the batch
    (context_fid.py:296-309)."""
    B = z1.size(0)
    if B == 1:
        return z1.new_tensor(0.0)
    z = torch.cat([z1, z2], dim=0).transpose(0, 1)          # T x 2B x C
    nll = _off_diagonal_nll(torch.matmul(z, z.transpose(1, 2)))
    i = torch.arange(B, device=z1.device)
    return (nll[:, i, B + i - 1].mean() + nll[:, B + i, i].mean()) / 2


def temporal_contrast(z1, z2):
    """Per series: the two views of a timestamp are the positive pair among the 2T timestamps of both views
    (context_fid.py:312-324)."""
    T = z1.size(1)
    if T == 1:
        return z1.new_tensor(0.0)
    z = torch.cat([z1, z2], dim=1)                          # B x 2T x C
    nll = _off_diagonal_nll(torch.matmul(z, z.transpose(1, 2)))
    t = torch.arange(T, device=z1.device)
    return (nll[:, t, T + t - 1].mean() + nll[:, T + t, t].mean()) / 2


def hierarchical_contrastive_loss(z1, z2, alpha=0.5, temporal_unit=0):
    """Mean over time scales (halved by max-pooling until one timestamp is left) of alpha * instance +
    (1 - alpha) * temporal contrast, the temporal term from scale `temporal_unit` on (context_fid.py:277-293)."""
    loss = torch.zeros((), device=z1.device)
    scales = 0
    while z1.size(1) > 1:
        if alpha != 0:
            loss = loss + alpha * instance_contrast(z1, z2)
        if scales >= temporal_unit and 1 - alpha != 0:
            loss = loss + (1 - alpha) * temporal_contrast(z1, z2)
        scales += 1
        z1 = F.max_pool1d(z1.transpose(1, 2), kernel_size=2).transpose(1, 2)
        z2 = F.max_pool1d(z2.transpose(1, 2), kernel_size=2).transpose(1, 2)
    if z1.size(1) == 1:
        if alpha != 0:
            loss = loss + alpha * instance_contrast(z1, z2)
        scales += 1
    return loss / scales


def _rows_from(x, start, length):
    """x[b, start[b] : start[b] + length] for every row b (context_fid.py:53-55)."""
    cols = start[:, None] + np.arange(length)
    return x[torch.arange(cols.shape[0])[:, None], cols]


def _split_long_series(data, max_len):
    """Series longer than 2 * max_len are cut along time into L // max_len near-equal sections, NaN-padded at the end
    to the first section's length and stacked as extra series (context_fid.py:31-50, :394-397)."""
    sections = data.shape[1] // max_len
    if sections < 2:
        return data
    parts = np.array_split(data, sections, axis=1)
    width = parts[0].shape[1]
    parts = [np.pad(p, [(0, 0), (0, width - p.shape[1]), (0, 0)], constant_values=np.nan) for p in parts]
    return np.concatenate(parts, axis=0)


def _centre_observed_span(x):
    """Rotate every series so that its all-NaN head and tail become equally long (context_fid.py:58-65)."""
    seen = ~np.isnan(x).all(axis=-1)
    head = np.argmax(seen, axis=1)
    tail = np.argmax(seen[:, ::-1], axis=1)
    shift = (head + tail) // 2 - head
    shift[shift < 0] += x.shape[1]
    rows, cols = np.ogrid[:x.shape[0], :x.shape[1]]
    return x[rows, cols - shift[:, None]]


class TS2VecFit:
    """The encoder, its contrastive fit and the full-series embedding `Context_FID` uses (context_fid.py:326-620)."""

    def __init__(self, input_dims, output_dims=320, hidden_dims=64, depth=10, device="cpu", lr=0.001, batch_size=16,
                 max_train_length=None, temporal_unit=0):
        self.device, self.lr, self.batch_size = device, lr, batch_size
        self.max_train_length, self.temporal_unit = max_train_length, temporal_unit
        self._net = SeriesEncoder(input_dims, output_dims, hidden_dims, depth).to(device)
        self.net = torch.optim.swa_utils.AveragedModel(self._net)  # equal-weight running mean of the iterates
        self.net.update_parameters(self._net)
        self.n_epochs = self.n_iters = 0

    def fit(self, train_data, n_epochs=None, n_iters=None):
        """train_data: numpy (n, T, C), NaN = missing.  Returns the mean loss of every completed epoch."""
        assert train_data.ndim == 3
        if n_iters is None and n_epochs is None:
            n_iters = 200 if train_data.size <= 100000 else 600
        if self.max_train_length is not None:
            train_data = _split_long_series(train_data, self.max_train_length)
        gap = np.isnan(train_data).all(axis=-1).any(axis=0)
        if gap[0] or gap[-1]:
            train_data = _centre_observed_span(train_data)
        train_data = train_data[~np.isnan(train_data).all(axis=2).all(axis=1)]
        data = TensorDataset(torch.from_numpy(train_data).to(torch.float))
        loader = DataLoader(data, batch_size=min(self.batch_size, len(data)), shuffle=True, drop_last=True)
        optimizer = torch.optim.AdamW(self._net.parameters(), lr=self.lr)
        epoch_losses = []
        while n_epochs is None or self.n_epochs < n_epochs:
            total, count, stop = 0.0, 0, False
            for (x,) in loader:
                if n_iters is not None and self.n_iters >= n_iters:
                    stop = True
                    break
                if self.max_train_length is not None and x.size(1) > self.max_train_length:
                    at = np.random.randint(x.size(1) - self.max_train_length + 1)
                    x = x[:, at:at + self.max_train_length]
                x = x.to(self.device)
                # two crops [eleft, right) and [left, eright) sharing [left, right), shifted per series by `offset`
                T = x.size(1)
                overlap = np.random.randint(low=2 ** (self.temporal_unit + 1), high=T + 1)
                left = np.random.randint(T - overlap + 1)
                right = left + overlap
                eleft = np.random.randint(left + 1)
                eright = np.random.randint(low=right, high=T + 1)
                offset = np.random.randint(low=-eleft, high=T - eright + 1, size=x.size(0))
                optimizer.zero_grad()
                out1 = self._net(_rows_from(x, offset + eleft, right - eleft))[:, -overlap:]
                out2 = self._net(_rows_from(x, offset + left, eright - left))[:, :overlap]
                loss = hierarchical_contrastive_loss(out1, out2, temporal_unit=self.temporal_unit)
                loss.backward()
                optimizer.step()
                self.net.update_parameters(self._net)
                total += loss.item()
                count += 1
                self.n_iters += 1
            if stop:
                break
            epoch_losses.append(total / count)
            self.n_epochs += 1
        return epoch_losses

    def encode_full_series(self, data, batch_size=None):
        """numpy (n, T, C) -> numpy (n, output_dims): the averaged encoder in eval mode, max over time
        (context_fid.py:480-488, :524-620 with encoding_window='full_series')."""
        assert data.ndim == 3
        was_training = self.net.training
        self.net.eval()
        loader = DataLoader(TensorDataset(torch.from_numpy(data).to(torch.float)),
                            batch_size=batch_size or self.batch_size)
        out = []
        with torch.no_grad():
            for (x,) in loader:
                r = self.net(x.to(self.device), None)
                out.append(F.max_pool1d(r.transpose(1, 2), kernel_size=r.size(1)).squeeze(-1).cpu())
        self.net.train(was_training)
        return torch.cat(out, dim=0).numpy()


def frechet_distance(act1, act2):
    """|mu1 - mu2|^2 + tr(S1 + S2 - 2 (S1 S2)^(1/2)) of two sets of activations (context_fid.py:640-653)."""
    mu1, cov1 = act1.mean(axis=0), np.cov(act1, rowvar=False)
    mu2, cov2 = act2.mean(axis=0), np.cov(act2, rowvar=False)
    root = scipy.linalg.sqrtm(cov1.dot(cov2))
    if np.iscomplexobj(root):
        root = root.real
    return np.sum((mu1 - mu2) ** 2.0) + np.trace(cov1 + cov2 - 2.0 * root)


def _to_numpy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def context_fid(ori_data, generated_data, device="cpu"):
    """One Context-FID evaluation (context_fid.py:655-665): fit on the real set, embed both, Frechet distance."""
    ori, gen = _to_numpy(ori_data), _to_numpy(generated_data)
    model = TS2VecFit(input_dims=ori.shape[-1], device=device, batch_size=8, lr=0.001, output_dims=320,
                      max_train_length=3000)
    model.fit(ori)
    ori_repr = model.encode_full_series(ori)
    gen_repr = model.encode_full_series(gen)
    order = np.random.permutation(ori.shape[0])
    return frechet_distance(ori_repr[order], gen_repr[order])


def summarise_scores(results):
    """(mean, sample std, 95 % confidence half-width), rounded to 4 places (context_fid.py:667-672); the last two
    are NaN for a single evaluation."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mean, std = np.mean(results), np.std(results, ddof=1)
        half = scipy.stats.sem(results) * scipy.stats.t.ppf((1 + 0.95) / 2.0, len(results) - 1)
    return round(float(mean), 4), round(float(std), 4), round(float(half), 4)


def calculate_fid(real_sig, gen_sig, device="cpu", iterations=1):
    """`iterations` evaluations on the first len(real_sig) generated series (context_fid.py:674-684)."""
    real, gen = _to_numpy(real_sig), _to_numpy(gen_sig)
    scores = [context_fid(real, gen[:real.shape[0]], device=device) for _ in range(iterations)]
    return summarise_scores(scores)
