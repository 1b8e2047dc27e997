"""The train iteration at the boundary between the TST completion stage and the diffusion step.

run_irregular.py:209-244 runs, per batch:

    h = embedder(x_ts, padding_masks); x_recon = decoder(h)         # :217-218  TST forward #1
    x_tilde_img = model.ts_to_img(x_recon)                          # :219      (carries grad_fn back into the TST)
    loss = model.loss_fn_irregular(x_tilde_img, mask) ... step      # :220-230  diffusion step
    h = embedder(x_ts, padding_masks); x_tilde = decoder(h)         # :233-234  TST forward #2 -- same weights as #1
    loss_e = sqrt(mse(x_tilde[obs], x_ts[obs])) * 10; step          # :235-243  TST's own step
    torch.cuda.empty_cache()                                        # :244      allocator flush every iteration

The TST stays on PyTorch-ROCm (BASELINE.json north_star); nothing of it is rewritten here.  What this module does is run
the SAME iteration without the three things that cost time and change no trained weight:

* the diffusion loss's gradient w.r.t. ``x_tilde_img`` is dead (only ``model.parameters()`` are in the diffusion
  optimiser, :117, and ``optimizer_er.zero_grad()`` wipes the TST gradients before they are used, :241 -- SURVEY §0.10):
  the image is detached, so no autograd graph is kept alive across the diffusion step and nothing is back-propagated
  into the TST from it;
* between forward #1 and forward #2 only the DIFFUSION weights change (:229): with ``reuse_completion`` the recovery
  loss is taken from forward #1's output (one TST forward + backward per iteration instead of two forwards + one
  backward).  Exact when the TST is deterministic; with its dropout live the two passes differ only in their masks;
* no ``empty_cache()``: the caching allocator keeps its blocks (the replayed diffusion graph owns its memory anyway).

``TrainIteration`` is the drop-in loop body; ``CompletionStandIn`` is a builder-written torch.nn network with the TST
stage's shapes (d_model 40, 5 heads, 6 layers, feed-forward 2048, 3-layer bi-GRU decoder: utils/utils_args.py:65-82,
run_irregular.py:143-163) used by tools/tst_boundary_bench.py and the tests -- it is NOT the reference's model and
is not meant to reproduce its outputs.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class CompletionStandIn(nn.Module):
    """Transformer encoder over (B, L, C) with a key-padding mask + 3-layer bidirectional GRU decoder + sigmoid:
    the layer types and sizes of the reference's completion stage, written from its configuration (see module doc)."""

    def __init__(self, feat_dim, max_len, d_model=40, n_heads=5, num_layers=6, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.d_model = d_model
        self.project_inp = nn.Linear(feat_dim, d_model)
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        self.register_buffer("pe", pe[None])
        layer = nn.TransformerEncoderLayer(d_model, n_heads, dim_feedforward, dropout, activation="gelu",
                                           batch_first=True)
        self.encoder = nn.TransformerEncoder(layer, num_layers, enable_nested_tensor=False)
        self.output_layer = nn.Linear(d_model, d_model)
        self.drop = nn.Dropout(dropout)
        hidden = int(d_model + (feat_dim - d_model) / 2)
        self.rnn = nn.GRU(input_size=d_model, hidden_size=hidden, bidirectional=True, num_layers=3, batch_first=True)
        self.linear = nn.Linear(hidden * 2, feat_dim)

    def embed(self, x, padding_masks):
        z = torch.nan_to_num(x, nan=0.0)
        z = self.drop(self.project_inp(z) * math.sqrt(self.d_model) + self.pe[:, : z.shape[1]])
        z = self.encoder(z, src_key_padding_mask=~padding_masks)
        return self.output_layer(self.drop(torch.nn.functional.gelu(z)))

    def decode(self, h):
        out, _ = self.rnn(h)
        return torch.sigmoid(self.linear(out))

    def forward(self, x, padding_masks):
        return self.decode(self.embed(x, padding_masks))


def recovery_loss(x_tilde, x_ts):
    """sqrt(mse over observed entries) * 10 (run_irregular.py:235-239 with _loss_e_t0 = MSE, _loss_e_0 = sqrt(.)*10)."""
    obs = ~torch.isnan(x_ts)
    return torch.sqrt(torch.nn.functional.mse_loss(x_tilde[obs], x_ts[obs])) * 10


def propagate_values(x_ts):
    """run_irregular.py:25-39 / :213 -- a series with EVERY entry missing is replaced, in place and in batch order, by
    its successor in the batch (the last one by its predecessor).  Without it such a series has an all-False padding
    mask: every key of the completion stage's attention is masked, its rows come out NaN, and the recovery loss and
    the completion weights with them.  Same result as the reference's per-sample loop -- a row is only ever replaced
    at its own turn, so looking the all-missing rows up once is the same thing -- for one host sync instead of one
    per sample."""
    B = x_ts.shape[0]
    for b in torch.isnan(x_ts).reshape(B, -1).all(dim=1).nonzero().flatten().tolist():
        x_ts[b] = x_ts[b + 1 if b + 1 < B else b - 1]
    return x_ts


class TrainIteration:
    """One iteration of run_irregular.py:209-244 for a diffusion model on the HIP path and a completion stage that stays
    on PyTorch-ROCm (``completion(x_ts, padding_masks) -> x_recon``; the reference's embedder + decoder composed, or any
    module of that signature).

    ``diffusion_step``: ``GraphedTrainStep`` (replayed HIP graph incl. clip + AdamW + EMA), or None for the eager
    sequence with ``diffusion_optimizer`` exactly as :220-230.  Returns ``(diffusion loss, recovery loss)`` as device
    scalars.  ``x_ts`` is the loader's batch without its time-index column (:211); like the reference's loop body the
    call starts with `propagate_values` on it (:213, in place; its one host sync is the only one in here --
    ``propagate=False`` for callers that already did it).

    ``reuse_completion`` changes how much dropout RNG the completion stage consumes (one forward instead of two), so
    with its dropout live the trajectory is a different, equally distributed one; ``reuse_completion=False`` keeps the
    reference's two forwards."""

    def __init__(self, model, completion, completion_optimizer, diffusion_step=None, diffusion_optimizer=None,
                 reuse_completion=True, empty_cache=False, propagate=True):
        if (diffusion_step is None) == (diffusion_optimizer is None):
            raise ValueError("give either a GraphedTrainStep or the diffusion optimiser of the eager loop")
        self.model, self.completion, self.opt_er = model, completion, completion_optimizer
        self.step, self.opt = diffusion_step, diffusion_optimizer
        self.reuse_completion = reuse_completion
        self.empty_cache = empty_cache
        self.propagate = propagate

    def __call__(self, x_ts):
        m = self.model
        if self.propagate:
            x_ts = propagate_values(x_ts)                                    # :213
        padding_masks = ~torch.isnan(x_ts).any(dim=-1)                       # :214
        x_img, mask = m.ts_to_img_with_mask(x_ts)                            # :215-216, one kernel
        x_recon = self.completion(x_ts, padding_masks)                       # :217-218
        x_tilde_img = m.ts_to_img(x_recon.detach())                          # :219, dead gradient cut
        if self.step is not None:
            loss = self.step(x_tilde_img, mask)                              # :220-230 as one replayed graph
        else:
            loss, _ = m.loss_fn_irregular(x_tilde_img, mask)
            self.opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            self.opt.step()
            m.on_train_batch_end()
            loss = loss.detach()
        x_tilde = x_recon if self.reuse_completion else self.completion(x_ts, padding_masks)  # :233-234
        loss_e = recovery_loss(x_tilde, x_ts)                                # :235-240
        self.opt_er.zero_grad()
        loss_e.backward()
        self.opt_er.step()                                                   # :241-243
        if self.empty_cache:
            torch.cuda.empty_cache()                                         # :244 (off: it only costs time)
        return loss, loss_e.detach()
