"""nn.Module call sites of BK-Core, mirrored so that a user of the reference finds the same constructor and
forward signatures on top of the HIP operator.

  * ``BKCoreLayer``           the BK-Core branch of ``MoEResNetBKLayer`` (reference src/models/resnet_bk.py:84-176,
                              216-219): v_proj -> tanh saturation -> he_diag -> BKCoreFunction (bf16 operands, complex64
                              device z, expand()ed off-diagonal buffers) -> clamp -> output_proj * bk_scale.  Parameter
                              and buffer names are the reference's, so its state_dict entries load unchanged.  The
                              FFN / MoE / attention branches of that layer are not part of this package.
  * ``CUDAOptimizedBKCore``   same constructor and ``forward(v) -> (B, N, 2)`` as the reference module of that name
                              (src/models/cuda_bk_core.py:243-413).  The reference class evaluates the un-regularised
                              recurrence in its own CUDA kernels and clamps Re/Im to +-10; here the numbers come from
                              the HIP operator (regularised, soft-clamped: the semantics of bk_core.py) followed by the
                              same +-10 clamp, so results agree to the regularisation's 1e-3 on well-conditioned input.
  * ``HybridGatedMixing``     what the reference's ``HybridHyperbolicAttention`` does with the resolvent diagonal
                              (src/models/phase7/hybrid_attention.py:84-86,110-147): soft caps of the two mixing
                              paths, the gate ``g = sigmoid(clamp(|Im G_ii|, max=100) * gate_sensitivity)``,
                              ``g * local + (1 - g) * global``, LayerNorm.  Same parameter names (``gate_sensitivity``,
                              ``layer_norm``).  The mixing paths themselves (windowed hyperbolic attention, AR-SSM) are
                              arguments: they are not part of this package.
  * ``BKHybridLayer``         ``MoEResNetBKLayer`` with ``use_hybrid_attention=True`` (resnet_bk.py:137-202), the
                              consumer configs/phase7_small.yaml:11 selects: the features are unused, ``G_ii`` feeds the
                              gate, and the operator's backward receives a cotangent on ``G_ii`` only.  The two mixing
                              paths are stand-ins (causal softmax attention in windows; a causal running mean over the
                              sequence through the library's cumulative sum).
  * ``LearnedTridiagonalGreen`` the Green-function step of the reference's hyperbolic BK-Core block
                              (src/models/phase8/bk_core_hyperbolic.py:282-286, 349-407): all three diagonals are
                              projections of the input (``he_diag_proj``, ``h0_super_proj``, ``h0_sub_proj``, same names and
                              initialisation), bf16 operands, ``BKCoreFunction.apply(he_diag, h0_super, h0_sub, z, False)``.
                              In the reference the off-diagonal projections never receive a gradient (the operator
                              returns None for them); with ``BKCoreFunction.GRAD_MODE = "analytic"`` they do.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .bk_core import BKCoreFunction, FusedBKLayerFunction, FusedSpectralOutputFunction
from .scan import fused_associative_scan


class BKCoreLayer(nn.Module):
    def __init__(self, d_model: int, n_seq: int, v_max: float = 5.0, feature_clamp: float | None = 5.0,
                 fused: bool = False):
        super().__init__()
        # fused: run the elementwise steps between v_proj and output_proj inside the library (two launches around the
        # operator instead of about a dozen tensor-library kernels per direction); same numbers as the unfused chain
        self.fused = fused
        self.d_model = d_model
        self.n_seq = n_seq
        self.v_proj = nn.Linear(d_model, 1)
        self.output_proj = nn.Linear(2, d_model)
        self.bk_scale = nn.Parameter(torch.ones(d_model, dtype=torch.float32))
        self.gamma = nn.Parameter(torch.tensor(0.0, dtype=torch.float32))
        self.epsilon_param = nn.Parameter(torch.tensor(1.0, dtype=torch.float32))
        with torch.no_grad():                                   # resnet_bk.py:92-95
            nn.init.normal_(self.v_proj.weight, mean=0.0, std=1e-4)
            nn.init.zeros_(self.v_proj.bias)
        # identity start: diagonal 1, off-diagonals 0 (resnet_bk.py:124-126)
        self.register_buffer("h0_diag_base", torch.full((1, n_seq), 1.0, dtype=torch.float32))
        self.register_buffer("h0_sub_base", torch.zeros((1, n_seq - 1), dtype=torch.float32))
        self.register_buffer("h0_super_base", torch.zeros((1, n_seq - 1), dtype=torch.float32))
        self.v_max = v_max
        self.feature_clamp = feature_clamp
        self.bk_core = BKCoreFunction.apply
        self.last_g_ii = None

    def spectral_features(self, x: torch.Tensor):
        """x (B, N, D) -> (features (B, N, 2) float32, G_ii (B, N, 1) complex64).  With ``fused`` the features are
        already clamped to +-feature_clamp."""
        B, N, _ = x.shape
        if N != self.n_seq:
            raise ValueError(f"Sequence length mismatch: expected {self.n_seq}, got {N}")
        v = self.v_proj(x).squeeze(-1)
        if self._use_fused():
            epsilon = F.softplus(self.epsilon_param) + 1e-4
            z = (1.0j * epsilon + 1j * self.gamma).to(dtype=torch.complex64, device=v.device)
            features, g = FusedBKLayerFunction.apply(v, self.h0_diag_base, self.h0_super_base, self.h0_sub_base, z,
                                                     self.v_max, self.feature_clamp)
            G_ii = g.unsqueeze(-1)          # the operator maps non-finite values to 0: nothing to replace
            self.last_g_ii = G_ii
            return features, G_ii
        v = torch.nan_to_num(v, nan=0.0, posinf=self.v_max, neginf=-self.v_max)
        v = self.v_max * torch.tanh(v / self.v_max)
        he_diag = self.h0_diag_base.expand(B, -1) + v
        epsilon = F.softplus(self.epsilon_param) + 1e-4
        z = (1.0j * epsilon + 1j * self.gamma).to(dtype=torch.complex64, device=he_diag.device)
        features, g = self.bk_core(he_diag.to(torch.bfloat16),
                                   self.h0_super_base.expand(B, -1).to(torch.bfloat16),
                                   self.h0_sub_base.expand(B, -1).to(torch.bfloat16), z)
        G_ii = g.unsqueeze(-1)
        bad = torch.isnan(G_ii.real) | torch.isnan(G_ii.imag)
        G_ii = torch.where(bad, torch.ones_like(G_ii), G_ii)    # no host sync: the replacement is unconditional
        self.last_g_ii = G_ii
        return features, G_ii

    def _use_fused(self) -> bool:
        return (self.fused and self.bk_core == BKCoreFunction.apply and BKCoreFunction.GRAD_MODE == "reference")

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None) -> torch.Tensor:
        """The layer's spectral branch: [residual +] bk_scale * output_proj(clamp(features)) (resnet_bk.py:216-219;
        ``residual`` is the reference's ``ffn_out``)."""
        if self._use_fused() and self.d_model <= 1024:
            # The clamped features never exist as a tensor: the operator writes G_ii alone, and the output stage reads
            # its float32 view through the clamp (forward and backward).
            B, N, _ = x.shape
            if N != self.n_seq:
                raise ValueError(f"Sequence length mismatch: expected {self.n_seq}, got {N}")
            v = self.v_proj(x).squeeze(-1)
            epsilon = F.softplus(self.epsilon_param) + 1e-4
            z = (1.0j * epsilon + 1j * self.gamma).to(dtype=torch.complex64, device=v.device)
            raw, g = FusedBKLayerFunction.apply(v, self.h0_diag_base, self.h0_super_base, self.h0_sub_base, z,
                                                self.v_max, None)
            self.last_g_ii = g.unsqueeze(-1)
            return FusedSpectralOutputFunction.apply(raw, self.output_proj.weight, self.output_proj.bias,
                                                     self.bk_scale, residual, self.feature_clamp)
        features, _ = self.spectral_features(x)
        if self._use_fused():                                          # clamped features; one launch for the rest
            if features.dtype == torch.float32 and self.d_model <= 1024:
                return FusedSpectralOutputFunction.apply(features, self.output_proj.weight, self.output_proj.bias,
                                                         self.bk_scale, residual)
        elif self.feature_clamp is not None:
            features = torch.clamp(features, -self.feature_clamp, self.feature_clamp)
        out = self.bk_scale * self.output_proj(features)
        return out if residual is None else residual + out


class HybridGatedMixing(nn.Module):
    def __init__(self, d_model: int, soft_cap: float = 30.0):
        super().__init__()
        self.gate_sensitivity = nn.Parameter(torch.tensor(1.0))            # hybrid_attention.py:52
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-5)                  # :55
        self.soft_cap = soft_cap                                           # :58

    def _cap(self, t: torch.Tensor) -> torch.Tensor:                       # :84-86, :110-112
        t = torch.tanh(t / self.soft_cap) * self.soft_cap
        return torch.nan_to_num(t, nan=0.0, posinf=self.soft_cap, neginf=-self.soft_cap)

    def forward(self, local_out: torch.Tensor, global_out: torch.Tensor, g_ii: torch.Tensor) -> torch.Tensor:
        """local_out, global_out (B, N, D): the two mixing paths before their soft caps, capped in the dtype they
        arrive in (bfloat16 under the reference's autocast); g_ii (B, N, 1) complex."""
        local_out, global_out = self._cap(local_out), self._cap(global_out)
        energy = torch.clamp(g_ii.imag.abs(), max=100.0)                   # :117-121
        g = torch.sigmoid(energy * self.gate_sensitivity)                  # :122
        self.last_gate_mean = g.detach().mean()
        out = g * local_out + (1 - g) * global_out                         # :141
        out = torch.nan_to_num(out, nan=0.0, posinf=self.soft_cap, neginf=-self.soft_cap)
        return self.layer_norm(out)                                        # :143


class BKHybridLayer(BKCoreLayer):
    """x -> v_proj -> operator -> G_ii -> gate between a local and a global mixing path of x.  ``output_proj`` and
    ``bk_scale`` exist (as in the reference layer, whose state_dict has them) and are unused on this path."""

    def __init__(self, d_model: int, n_seq: int, num_heads: int = 8, window: int = 32, rank: int = 16,
                 v_max: float = 5.0, fused: bool = False):
        super().__init__(d_model, n_seq, v_max=v_max, feature_clamp=5.0, fused=fused)
        if d_model % num_heads:
            raise ValueError("d_model must be a multiple of num_heads")
        self.num_heads, self.window = num_heads, window
        self.hybrid_attn = HybridGatedMixing(d_model)
        # stand-ins for HyperbolicMultiHeadAttention (windowed, causal) and AdaptiveRankSemiseparableLayer
        self.local_qkv = nn.Linear(d_model, 3 * d_model, bias=False)
        self.local_out = nn.Linear(d_model, d_model, bias=False)
        self.global_in = nn.Linear(d_model, rank, bias=False)
        self.global_out = nn.Linear(rank, d_model)

    def local_path(self, x: torch.Tensor) -> torch.Tensor:
        B, N, D = x.shape
        W = self.window
        pad = (W - N % W) % W                                              # hybrid_attention.py:93-100
        xp = F.pad(x, (0, 0, 0, pad)) if pad else x
        nw = xp.shape[1] // W
        q, k, v = self.local_qkv(xp).reshape(B * nw, W, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True)        # causal inside each window (:103)
        return self.local_out(o.transpose(1, 2).reshape(B, nw * W, D))[:, :N]

    def global_path(self, x: torch.Tensor) -> torch.Tensor:
        u = self.global_in(x)                                              # (B, N, rank)
        csum = fused_associative_scan(u, dim=1) if u.is_cuda else torch.cumsum(u, dim=1)   # ar_ssm_layer.py:517-525
        pos = torch.arange(1, x.shape[1] + 1, device=x.device, dtype=u.dtype)
        return self.global_out(csum / pos[None, :, None])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _, G_ii = self.spectral_features(x)                                # the features are not used on this path
        return self.hybrid_attn(self.local_path(x), self.global_path(x), G_ii)


class CUDAOptimizedBKCore(nn.Module):
    def __init__(self, n_seq: int, use_cuda_kernels: bool = True):
        super().__init__()
        self.n_seq = n_seq
        self.use_cuda_kernels = use_cuda_kernels                # kept; the HIP kernels are the only back-end
        self.cuda_available = True
        self.register_buffer("h0_diag", torch.full((n_seq,), -2.0))
        self.register_buffer("h0_sub", torch.full((n_seq - 1,), 1.0))
        self.register_buffer("h0_super", torch.full((n_seq - 1,), 1.0))
        self.z = torch.tensor(1.0j, dtype=torch.complex128)

    def forward(self, v: torch.Tensor) -> torch.Tensor:
        B, N = v.shape
        he_diag = self.h0_diag.unsqueeze(0).expand(B, -1) + v
        features, _ = BKCoreFunction.apply(he_diag.float(), self.h0_super.unsqueeze(0).expand(B, -1),
                                           self.h0_sub.unsqueeze(0).expand(B, -1), complex(self.z))
        return torch.clamp(features, -10.0, 10.0)


class LearnedTridiagonalGreen(nn.Module):
    def __init__(self, d_model: int, z_real: float = 0.0, z_imag: float = 0.1):
        super().__init__()
        self.he_diag_proj = nn.Linear(d_model, d_model)
        self.h0_super_proj = nn.Linear(d_model, d_model)
        self.h0_sub_proj = nn.Linear(d_model, d_model)
        self.register_buffer("z", torch.complex(torch.tensor(z_real), torch.tensor(z_imag)))
        # identity start: diagonal 1, off-diagonals 0, weights ~ N(0, 1e-4) (bk_core_hyperbolic.py:349-365)
        nn.init.normal_(self.he_diag_proj.weight, mean=0.0, std=1e-4)
        nn.init.constant_(self.he_diag_proj.bias, 1.0)
        for proj in (self.h0_super_proj, self.h0_sub_proj):
            nn.init.normal_(proj.weight, mean=0.0, std=1e-4)
            nn.init.zeros_(proj.bias)

    def compute_green_function(self, x: torch.Tensor):
        """x (B, N, D) -> (G_ii (B, N) complex64, features (B, N, 2) float32)."""
        xb = x.to(torch.bfloat16)
        w = lambda proj: (proj.weight.to(torch.bfloat16), proj.bias.to(torch.bfloat16))
        he_diag = F.linear(xb, *w(self.he_diag_proj)).mean(dim=-1)
        h0_super = F.linear(xb[:, :-1], *w(self.h0_super_proj)).mean(dim=-1)
        h0_sub = F.linear(xb[:, 1:], *w(self.h0_sub_proj)).mean(dim=-1)
        features, G_ii = BKCoreFunction.apply(he_diag, h0_super, h0_sub, self.z.to(x.device), False)
        return G_ii, features

    def forward(self, x: torch.Tensor):
        return self.compute_green_function(x)
