"""ChiTransformer -- Diffusion Policy's transformer denoiser: action tokens decode against a memory of
[timestep token | observation tokens] with a causal target mask and a staggered memory mask
(interface/checkpoint contract: reference nn_diffusion/chitransformer.py:12-158).

Execution: on a ROCm device without autograd, forward and every sample() over it run through engine/bigbatch.py ->
``cdx_chitf_run`` (memory tokens and their K/V projections precomputed per request, masked MFMA self-attention, small
cross-attention kernel, GEMMs with fused GELU / residual epilogues); WITH autograd (loss() / update()) through
engine/train.py:chitf_forward (Linear / LayerNorm / attention nodes on the library's kernels, forward and backward, attention dropout
included); the module code below is the CPU executor.
"""
from typing import Optional

import torch
import torch.nn as nn

from ..utils import FourierEmbedding, PositionalEmbedding, SinusoidalEmbedding
from .base_nn_diffusion import BaseNNDiffusion

_PASSIVE = (nn.Dropout, SinusoidalEmbedding, FourierEmbedding, PositionalEmbedding, nn.TransformerEncoderLayer,
            nn.TransformerDecoderLayer, nn.TransformerEncoder, nn.TransformerDecoder, nn.ModuleList, nn.Mish,
            nn.Sequential)


def init_weight(module):
    """GPT-style init: N(0, 0.02) weights, zero biases, unit LayerNorm (reference chitransformer.py:12-58)."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
        if isinstance(module, nn.Linear) and module.bias is not None:
            torch.nn.init.zeros_(module.bias)
    elif isinstance(module, nn.MultiheadAttention):
        for name in ("in_proj_weight", "q_proj_weight", "k_proj_weight", "v_proj_weight"):
            w = getattr(module, name)
            if w is not None:
                torch.nn.init.normal_(w, mean=0.0, std=0.02)
        for name in ("in_proj_bias", "bias_k", "bias_v"):
            b = getattr(module, name)
            if b is not None:
                torch.nn.init.zeros_(b)
    elif isinstance(module, nn.LayerNorm):
        torch.nn.init.zeros_(module.bias)
        torch.nn.init.ones_(module.weight)
    elif isinstance(module, ChiTransformer):
        torch.nn.init.normal_(module.pos_emb, mean=0.0, std=0.02)
        if module.obs_emb is not None:
            torch.nn.init.normal_(module.cond_pos_emb, mean=0.0, std=0.02)
    elif not isinstance(module, _PASSIVE):
        raise RuntimeError("Unaccounted module {}".format(module))


def _additive_mask(allowed: torch.Tensor) -> torch.Tensor:
    m = allowed.float()
    return m.masked_fill(allowed == 0, float("-inf")).masked_fill(allowed == 1, float(0.0))


class ChiTransformer(BaseNNDiffusion):
    def __init__(self, act_dim: int, obs_dim: int, Ta: int, To: int, d_model: int = 256, nhead: int = 4,
                 num_layers: int = 8, p_drop_emb: float = 0.0, p_drop_attn: float = 0.3, n_cond_layers: int = 0,
                 timestep_emb_type: str = "positional", timestep_emb_params: Optional[dict] = None):
        super().__init__(d_model, timestep_emb_type, timestep_emb_params)
        self.To, self.obs_dim, self.T, self.T_cond = To, obs_dim, Ta, 1 + To
        self.act_emb = nn.Linear(act_dim, d_model)
        self.pos_emb = nn.Parameter(torch.zeros(1, Ta, d_model))
        self.obs_emb = nn.Linear(obs_dim, d_model)
        self.cond_pos_emb = nn.Parameter(torch.zeros(1, 1 + To, d_model))
        self.drop = nn.Dropout(p_drop_emb)

        def ffn():
            return nn.Sequential(nn.Linear(d_model, 4 * d_model), nn.Mish(), nn.Linear(4 * d_model, d_model))

        self.cond_encoder = ffn()
        if n_cond_layers > 0:
            layer = nn.TransformerEncoderLayer(d_model, nhead, 4 * d_model, p_drop_attn, activation="gelu",
                                               batch_first=True, norm_first=True)
            self.encoder = nn.TransformerEncoder(encoder_layer=layer, num_layers=n_cond_layers)
        else:
            self.encoder = ffn()
        dec = nn.TransformerDecoderLayer(d_model, nhead, 4 * d_model, p_drop_attn, activation="gelu",
                                         batch_first=True, norm_first=True)
        self.decoder = nn.TransformerDecoder(decoder_layer=dec, num_layers=num_layers)

        causal = (torch.triu(torch.ones(Ta, Ta)) == 1).transpose(0, 1)          # token i sees tokens <= i
        self.mask = nn.Parameter(_additive_mask(causal), requires_grad=False)
        t, s = torch.meshgrid(torch.arange(Ta), torch.arange(To + 1), indexing="ij")
        self.memory_mask = nn.Parameter(_additive_mask(t >= (s - 1)), requires_grad=False)
        self.ln_f = nn.LayerNorm(d_model)
        self.head = nn.Linear(d_model, act_dim)
        self.apply(init_weight)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, Ta, act_dim), noise (b,), condition (b, To, obs_dim)|None(=zeros) -> (b, Ta, act_dim)."""
        from ..engine import dispatch, train
        if train.supports_chitf(self, x, condition):
            return train.chitf_forward(self, x, noise, condition)           # autograd on, ROCm device: loss() / update()
        y = dispatch.try_backbone_forward(self, x, noise, condition)        # cdx_chitf_run on a ROCm device
        if y is not None:
            return y
        return self._forward_torch(x, noise, condition)

    def _forward_torch(self, x, noise, condition=None):
        if condition is None:
            condition = torch.zeros((x.shape[0], self.To, self.obs_dim)).to(x.device)
        cond_tok = torch.cat([self.map_noise(noise).unsqueeze(1), self.obs_emb(condition)], dim=1)
        memory = self.encoder(self.drop(cond_tok + self.cond_pos_emb[:, :cond_tok.shape[1], :]))
        act_tok = self.act_emb(x)
        h = self.drop(act_tok + self.pos_emb[:, :act_tok.shape[1], :])
        h = self.decoder(tgt=h, memory=memory, tgt_mask=self.mask, memory_mask=self.memory_mask)
        return self.head(self.ln_f(h))
