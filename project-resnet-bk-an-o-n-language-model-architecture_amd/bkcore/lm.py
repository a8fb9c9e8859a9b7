"""A small ResNet-BK language model around the BK-Core operator, for the end-to-end training-step configuration
(phase7_small dimensions of the reference: d_model 256, 4 layers, n_seq 256, vocab 50257, configs/phase7_small.yaml:7-33).

This is a plain restatement of the classic model's data flow (reference src/models/resnet_bk.py:350-410, 227-300:
token + position embedding, pre-norm residual blocks whose mixing path is the BK-Core spectral branch plus a feed-forward
network, final LayerNorm, vocabulary head).  ``hybrid=True`` selects what configs/phase7_small.yaml:11 selects in the
reference (``use_hybrid_attention``): the block's mixing path is gated by the resolvent diagonal
(``bkcore.layers.BKHybridLayer``: the reference's gate, resnet_bk.py:172-185 and hybrid_attention.py:117-143, around
stand-in local and global paths), the features go unused and the operator's backward sees a cotangent on G_ii alone.
MoE routing and the HTT embedding are not part of it.  What it exercises is the operator inside a real backward pass and, with more than one
process, the data-parallel gradient all-reduce around it.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import BKCoreLayer, BKHybridLayer


class ResNetBKBlock(nn.Module):
    """``ffn_mult = 2, ffn_activation = "relu"`` is the reference's feed-forward branch with a single expert
    (src/models/moe.py:61-69 with num_experts = 1: the softmax over one gate is 1, so the layer is its one expert)."""

    def __init__(self, d_model: int, n_seq: int, ffn_mult: int = 4, fused: bool = False, ffn_activation: str = "gelu",
                 hybrid: bool = False):
        super().__init__()
        if ffn_activation not in ("gelu", "relu"):
            raise ValueError(f"unknown ffn_activation {ffn_activation!r}")
        self.layer_norm = nn.LayerNorm(d_model)
        self.hybrid = hybrid
        if hybrid:      # no feed-forward branch: the reference builds none for a hybrid layer (resnet_bk.py:61-82)
            self.bk_layer = BKHybridLayer(d_model, n_seq, fused=fused)
            self.ffn = None
            return
        self.bk_layer = BKCoreLayer(d_model, n_seq, fused=fused)
        act = nn.GELU() if ffn_activation == "gelu" else nn.ReLU()
        self.ffn = nn.Sequential(nn.Linear(d_model, ffn_mult * d_model), act, nn.Linear(ffn_mult * d_model, d_model))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = self.layer_norm(x)
        if self.hybrid:
            return x + self.bk_layer(h)                                    # resnet_bk.py:243-251
        if self.bk_layer.fused:          # ffn_out + bk_scale * spec_out in the layer's output launch (resnet_bk.py:219)
            return x + self.bk_layer(h, residual=self.ffn(h))
        return x + self.bk_layer(h) + self.ffn(h)


class SmallResNetBKLM(nn.Module):
    def __init__(self, vocab_size: int = 50257, d_model: int = 256, n_layers: int = 4, n_seq: int = 256,
                 fused: bool = False, ffn_mult: int = 4, ffn_activation: str = "gelu", hybrid: bool = False):
        super().__init__()
        self.n_seq = n_seq
        self.token_embedding = nn.Embedding(vocab_size, d_model)
        self.position_embedding = nn.Embedding(n_seq, d_model)
        self.blocks = nn.ModuleList([ResNetBKBlock(d_model, n_seq, ffn_mult=ffn_mult, fused=fused,
                                                   ffn_activation=ffn_activation, hybrid=hybrid)
                                     for _ in range(n_layers)])
        self.layer_norm_final = nn.LayerNorm(d_model)
        self.lm_head = nn.Linear(d_model, vocab_size)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.position_embedding.weight, std=0.02)

    def set_bk_core(self, fn) -> None:
        """Replace the operator callable of every block (tests plug a CPU oracle in here)."""
        for blk in self.blocks:
            blk.bk_layer.bk_core = fn

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        B, N = tokens.shape
        pos = torch.arange(N, device=tokens.device)
        x = self.token_embedding(tokens) + self.position_embedding(pos)[None]
        x = torch.clamp(torch.nan_to_num(x, nan=0.0, posinf=10.0, neginf=-10.0), -10.0, 10.0)  # resnet_bk.py:469-470
        for blk in self.blocks:
            x = blk(x)
        return self.lm_head(self.layer_norm_final(x))


def training_step(model: nn.Module, optimizer: torch.optim.Optimizer, tokens: torch.Tensor, targets: torch.Tensor,
                  grad_clip: float = 1.0) -> torch.Tensor:
    """One optimisation step as the reference trainer does it (scripts/train_phase7.py:397-440: cross-entropy over
    all positions, backward, gradient-norm clip, AdamW step).  `model` may be wrapped in DistributedDataParallel:
    the gradient all-reduce (RCCL on GPUs) happens inside backward().  Returns the detached loss."""
    optimizer.zero_grad(set_to_none=True)
    logits = model(tokens)
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1))
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
    optimizer.step()
    return loss.detach()
