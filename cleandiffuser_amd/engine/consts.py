"""Constants and weight-layout helpers shared by the program compiler (``program2.py``), the GEMM building blocks (``blocks.py``)
and the host side of the executors.

Activation ids mirror ``CDX_ACT_*`` of ``include/cdx.h``; ``MODE_*`` / ``GN_EPS`` mirror ``csrc/cdx_ops2.h``
(``tests/test_abi_contract.py`` compares).  Conv weights are handed to the packers as ``[C_out][tap][C_in]`` ("effective" layout):
a ``Conv1d`` as it is, a ``ConvTranspose1d`` as the equivalent gather.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

MODE_16X16, MODE_4X4 = 0, 1        # MFMA shape of a conv op's records: 16x16x4 (16 rows x 16 cols x 16 K per record) or
                                   # 4x4x1 x16 blocks (64 rows x 4 cols x 4 K per record) for <= 8 positions
ACT_NONE, ACT_MISH, ACT_GELU_ERF, ACT_LEAKY, ACT_SILU, ACT_RELU, ACT_GELU_TANH, ACT_MISH_GRAD, ACT_TANH = range(9)
GN_EPS = 1e-5
MLP_TILE = 16                      # largest tile (samples per workgroup) of the batch-tiled MLP programs


def _fbits(x: float) -> int:
    """fp32 bit pattern as a (signed) int32 op word."""
    return int(np.float32(x).view(np.int32))


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


def slot_stride(c: int) -> int:
    """Row stride (floats) of a channel-last LDS slot: the +4 keeps ``ds_read_b128`` of 16 different rows on different banks."""
    return pad16(c) + 4


def _conv1d_eff(conv: nn.Conv1d) -> torch.Tensor:
    return conv.weight.detach().permute(0, 2, 1)            # (C_out, C_in, k) -> [co][tap][ci]


def _convT1d_eff(conv: nn.ConvTranspose1d) -> torch.Tensor:
    return conv.weight.detach().permute(1, 2, 0)            # (C_in, C_out, k) -> [co][tap][ci]


def supports_janner(net) -> Optional[str]:
    """None if the fused kernel can run this JannerUNet1d, else the reason it cannot."""
    if net.attention:
        return "attention=True (LinearAttention) is PyTorch-only"
    if net.norm_type != "groupnorm":
        return f"norm_type={net.norm_type!r} is PyTorch-only"
    if net.kernel_size % 2 == 0:
        return f"kernel_size={net.kernel_size} unsupported (odd only)"
    return None


def _act_id(m) -> Optional[int]:
    """nn activation module -> ACT_* id of the program kernel, None when it has no native epilogue."""
    if isinstance(m, nn.ReLU):
        return ACT_RELU
    if isinstance(m, nn.Mish):
        return ACT_MISH
    if isinstance(m, nn.SiLU):
        return ACT_SILU
    if isinstance(m, nn.Tanh):
        return ACT_TANH
    if isinstance(m, nn.GELU):
        return ACT_GELU_TANH if getattr(m, "approximate", "none") == "tanh" else ACT_GELU_ERF
    if isinstance(m, nn.LeakyReLU) and abs(m.negative_slope - 0.01) < 1e-12:
        return ACT_LEAKY
    if isinstance(m, nn.Identity):
        return ACT_NONE
    return None
