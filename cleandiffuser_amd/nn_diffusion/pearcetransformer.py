"""PearceTransformer -- the token transformer of "Imitating Human Behaviour with Diffusion Models" (interface/checkpoint
contract: reference nn_diffusion/pearcetransformer.py:8-151).

Tokens: [action embedding | timestep embedding | To observation embeddings], each projected to ``trans_emb_dim`` and tagged
with a sine positional code of its index (``TimeSiren``); four encoder blocks (Linear -> qkv, seq-first
``nn.MultiheadAttention`` over ``trans_emb_dim * nhead`` features, 1/1.414-scaled residuals, BatchNorm1d over the feature
axis -- running statistics in eval mode); the flattened tokens feed one Linear head.  On a ROCm device sampling and eval-mode
forwards run on ``cdx_pearcetf_run`` (engine/bigbatch.py folds everything linear); training stays on autograd.
"""
from typing import Optional

import torch
import torch.nn as nn

from .base_nn_diffusion import BaseNNDiffusion

_S = 1.414


class TimeSiren(nn.Module):
    def __init__(self, input_dim, emb_dim):
        super().__init__()
        self.lin1 = nn.Linear(input_dim, emb_dim, bias=False)
        self.lin2 = nn.Linear(emb_dim, emb_dim)

    def forward(self, x):
        return self.lin2(torch.sin(self.lin1(x)))


class TransformerEncoderBlock(nn.Module):
    def __init__(self, trans_emb_dim, transformer_dim, nheads):
        super().__init__()
        self.trans_emb_dim, self.transformer_dim, self.nheads = trans_emb_dim, transformer_dim, nheads
        self.input_to_qkv1 = nn.Linear(trans_emb_dim, transformer_dim * 3)
        self.multihead_attn1 = nn.MultiheadAttention(transformer_dim, num_heads=nheads)
        self.attn1_to_fcn = nn.Linear(transformer_dim, trans_emb_dim)
        self.attn1_fcn = nn.Sequential(nn.Linear(trans_emb_dim, trans_emb_dim * 4), nn.GELU(),
                                       nn.Linear(trans_emb_dim * 4, trans_emb_dim))
        self.norm1a = nn.BatchNorm1d(trans_emb_dim)
        self.norm1b = nn.BatchNorm1d(trans_emb_dim)

    def split_qkv(self, qkv):
        assert qkv.shape[-1] == self.transformer_dim * 3
        return qkv.split(self.transformer_dim, dim=-1)

    @staticmethod
    def _feature_norm(norm, seq_first):            # (tokens, batch, feat) -> BatchNorm1d over feat with (batch, feat, tokens)
        return norm(seq_first.permute(1, 2, 0)).permute(2, 0, 1)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        q, k, v = self.split_qkv(self.input_to_qkv1(inputs))
        att = self.multihead_attn1(q, k, v, need_weights=False)[0]
        h = self._feature_norm(self.norm1a, self.attn1_to_fcn(att) / _S + inputs / _S)
        return self._feature_norm(self.norm1b, self.attn1_fcn(h) / _S + h / _S)


class EmbeddingBlock(nn.Module):
    def __init__(self, in_dim: int, emb_dim: int):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(in_dim, emb_dim), nn.LeakyReLU(), nn.Linear(emb_dim, emb_dim))

    def forward(self, x):
        return self.model(x)


class PearceTransformer(BaseNNDiffusion):
    def __init__(self, act_dim: int, To: int = 1, emb_dim: int = 128, trans_emb_dim: int = 64, nhead: int = 16,
                 timestep_emb_type: str = "positional", timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.To, self.emb_dim = To, emb_dim
        self.act_emb = nn.Sequential(nn.Linear(act_dim, emb_dim), nn.LeakyReLU(), nn.Linear(emb_dim, emb_dim))
        width = trans_emb_dim * nhead
        self.act_to_input = nn.Linear(emb_dim, trans_emb_dim)
        self.t_to_input = nn.Linear(emb_dim, trans_emb_dim)
        self.cond_to_input = nn.Linear(emb_dim, trans_emb_dim)
        self.pos_embed = TimeSiren(1, trans_emb_dim)
        self.transformer_blocks = nn.Sequential(*[TransformerEncoderBlock(trans_emb_dim, width, nhead) for _ in range(4)])
        self.final = nn.Linear(trans_emb_dim * (2 + To), act_dim)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, act_dim), noise (b,), condition (b, To, emb_dim)|None(=zeros) -> (b, act_dim)."""
        if type(self) is PearceTransformer:
            from ..engine import dispatch
            y = dispatch.try_backbone_forward(self, x, noise, condition)     # cdx_pearcetf_run on a ROCm device (eval mode)
            if y is not None:
                return y
        if condition is None:
            condition = torch.zeros((x.shape[0], self.To, self.emb_dim), device=x.device)
        dev = x.device

        def pos(idx):
            return self.pos_embed(idx.to(dev, torch.float32))

        tok_x = self.act_to_input(self.act_emb(x)) + pos(torch.full((1, 1), 1.0))
        tok_t = self.t_to_input(self.map_noise(noise)) + pos(torch.full((1, 1), 2.0))
        tok_c = self.cond_to_input(condition) + pos(torch.arange(3, 3 + condition.shape[1])[None, :, None])
        tokens = torch.cat([tok_x.unsqueeze(1), tok_t.unsqueeze(1), tok_c], dim=1)
        tokens = self.transformer_blocks(tokens.permute(1, 0, 2)).permute(1, 0, 2)
        return self.final(torch.flatten(tokens, start_dim=1))
