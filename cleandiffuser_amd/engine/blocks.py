"""ctypes front end of the big-batch building blocks of libcdx.so (include/cdx.h: cdx_gemm_f32, cdx_layernorm_f32,
cdx_attention_f32, cdx_act_f32).  Every tensor crosses as a raw device pointer; outputs are caller-allocated.
These are used when M = batch x tokens >> 256 (DiT1d, wide MLPs), where the layers are classic GEMMs."""
import ctypes
import os
from typing import Optional

import torch

from . import consts as P
from .runtime import _check, _stream_ptr, load_library

ACT = {"none": P.ACT_NONE, "mish": P.ACT_MISH, "gelu": P.ACT_GELU_ERF, "leaky": P.ACT_LEAKY, "silu": P.ACT_SILU,
       "relu": P.ACT_RELU, "gelu_tanh": P.ACT_GELU_TANH, "mish_grad": P.ACT_MISH_GRAD, "tanh": P.ACT_TANH}


class CdxGemmArgs(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("gate", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("table", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("lda", ctypes.c_int32),
                ("ldw", ctypes.c_int32), ("ldc", ctypes.c_int32), ("ldg", ctypes.c_int32), ("ldr", ctypes.c_int32),
                ("rows_per_gate", ctypes.c_int32), ("table_rows", ctypes.c_int32), ("act", ctypes.c_int32),
                ("conv_taps", ctypes.c_int32), ("conv_cin", ctypes.c_int32), ("conv_lin", ctypes.c_int32),
                ("conv_lout", ctypes.c_int32), ("conv_stride", ctypes.c_int32), ("conv_pad", ctypes.c_int32),
                ("partial", ctypes.c_void_p), ("partial_slices", ctypes.c_int32)]


class CdxGnArgs(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("fa", ctypes.c_void_p), ("fb", ctypes.c_void_p), ("residual", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("B", "L", "C", "G", "ldx", "ldy", "ldr", "ldfa", "ldfb", "fa_row", "fa_per_sample",
                                              "film_mode", "act")] + [("eps", ctypes.c_float)] + \
               [("dgamma_part", ctypes.c_void_p), ("dbeta_part", ctypes.c_void_p), ("dgamma_sum", ctypes.c_void_p), ("dbeta_sum", ctypes.c_void_p),
                ("dy_possum", ctypes.c_void_p), ("ld_possum", ctypes.c_int32)]


class CdxWgradArgs(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("q", ctypes.c_void_p), ("dw", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("batch", "l_p", "l_q", "ca", "cb", "taps", "stride", "pad", "ldp", "ldq", "k_split")] + \
               [("db", ctypes.c_void_p)]


WGRAD_BATCH = 32


class CdxWgradBatch(ctypes.Structure):
    _fields_ = [("n_jobs", ctypes.c_int32), ("wg_start", ctypes.c_int32 * (WGRAD_BATCH + 1)), ("job", CdxWgradArgs * WGRAD_BATCH)]


class CdxRelayoutJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p)] + [(n, ctypes.c_int32) for n in ("n0", "n1", "n2", "s0", "s1", "s2")]


RELAYOUT_CHUNK = 2048
GATHER_MAX_FIELDS = 8


class CdxGatherField(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("out", ctypes.c_void_p), ("width", ctypes.c_int32), ("steps", ctypes.c_int32)]


class CdxGatherArgs(ctypes.Structure):
    _fields_ = [("row0", ctypes.c_void_p), ("batch", ctypes.c_int32), ("n_fields", ctypes.c_int32), ("rows", ctypes.c_longlong),
                ("field", CdxGatherField * GATHER_MAX_FIELDS)]


class CdxLnArgs(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("C", ctypes.c_int32), ("ldx", ctypes.c_int32), ("ldy", ctypes.c_int32),
                ("ldmod", ctypes.c_int32), ("rows_per_mod", ctypes.c_int32), ("eps", ctypes.c_float),
                ("x_rows", ctypes.c_int32)]


class CdxLnBwdArgs(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("dyxhat", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("scale", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("M", "C", "ldx", "lddy", "lddx", "ldmod", "rows_per_mod")] + [("eps", ctypes.c_float)]


class CdxAttnBwdArgs(ctypes.Structure):
    _fields_ = [("qkv", ctypes.c_void_p), ("dout", ctypes.c_void_p), ("dqkv", ctypes.c_void_p), ("B", ctypes.c_int32), ("T", ctypes.c_int32),
                ("n_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("scale", ctypes.c_float)]


class CdxMhaTrainArgs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("q", "k", "v", "mask", "keep", "out", "dout", "dq", "dk", "dv")] + \
               [(n, ctypes.c_int32) for n in ("B", "Tq", "Tk", "n_heads", "head_dim", "ldq", "ldk", "ldv", "ldo", "lddq", "lddk", "lddv")] + \
               [("scale", ctypes.c_float)]


class CdxAttnArgs(ctypes.Structure):
    _fields_ = [("qkv", ctypes.c_void_p), ("out", ctypes.c_void_p), ("B", ctypes.c_int32), ("T", ctypes.c_int32),
                ("n_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("scale", ctypes.c_float),
                ("mask", ctypes.c_void_p)]


class CdxXattnArgs(ctypes.Structure):
    _fields_ = [("q", ctypes.c_void_p), ("kv_shared", ctypes.c_void_p), ("kv_rows", ctypes.c_void_p), ("mask", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("B", ctypes.c_int32), ("T", ctypes.c_int32), ("n_obs", ctypes.c_int32),
                ("n_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("shared_row", ctypes.c_int32),
                ("shared_per_sample", ctypes.c_int32), ("scale", ctypes.c_float)]


_declared = False


def _lib():
    global _declared
    lib = load_library()
    if not _declared:
        lib.cdx_gemm_f32.argtypes = [ctypes.POINTER(CdxGemmArgs), ctypes.c_void_p]
        lib.cdx_layernorm_f32.argtypes = [ctypes.POINTER(CdxLnArgs), ctypes.c_void_p]
        lib.cdx_attention_f32.argtypes = [ctypes.POINTER(CdxAttnArgs), ctypes.c_void_p]
        lib.cdx_groupnorm_f32.argtypes = [ctypes.POINTER(CdxGnArgs), ctypes.c_void_p]
        lib.cdx_groupnorm_f32.restype = ctypes.c_int
        lib.cdx_groupnorm_bwd_f32.argtypes = [ctypes.POINTER(CdxGnArgs), ctypes.c_void_p]
        lib.cdx_groupnorm_bwd_f32.restype = ctypes.c_int
        lib.cdx_cross_attention_f32.argtypes = [ctypes.POINTER(CdxXattnArgs), ctypes.c_void_p]
        lib.cdx_cross_attention_f32.restype = ctypes.c_int
        lib.cdx_act_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
        lib.cdx_act_bwd_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_void_p]
        lib.cdx_act_bwd_f32.restype = ctypes.c_int
        lib.cdx_conv_wgrad_f32.argtypes = [ctypes.POINTER(CdxWgradArgs), ctypes.c_void_p]
        lib.cdx_conv_wgrad_f32.restype = ctypes.c_int
        lib.cdx_conv_wgrad_batch_f32.argtypes = [ctypes.POINTER(CdxWgradBatch), ctypes.c_void_p]
        lib.cdx_conv_wgrad_batch_f32.restype = ctypes.c_int
        lib.cdx_colsum_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        lib.cdx_colsum_f32.restype = ctypes.c_int
        lib.cdx_layernorm_bwd_f32.argtypes = [ctypes.POINTER(CdxLnBwdArgs), ctypes.c_void_p]
        lib.cdx_layernorm_bwd_f32.restype = ctypes.c_int
        lib.cdx_attention_bwd_f32.argtypes = [ctypes.POINTER(CdxAttnBwdArgs), ctypes.c_void_p]
        lib.cdx_attention_bwd_f32.restype = ctypes.c_int
        for f in (lib.cdx_mha_train_fwd_f32, lib.cdx_mha_train_bwd_f32):
            f.argtypes = [ctypes.POINTER(CdxMhaTrainArgs), ctypes.c_void_p]
            f.restype = ctypes.c_int
        lib.cdx_relayout_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        lib.cdx_relayout_f32.restype = ctypes.c_int
        lib.cdx_gather_windows_f32.argtypes = [ctypes.POINTER(CdxGatherArgs), ctypes.c_void_p]
        lib.cdx_gather_windows_f32.restype = ctypes.c_int
        for f in (lib.cdx_gemm_f32, lib.cdx_layernorm_f32, lib.cdx_attention_f32, lib.cdx_act_f32):
            f.restype = ctypes.c_int
        _declared = True
    return lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _rows(t: torch.Tensor):
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1) and t.dtype == torch.float32 and t.is_cuda, \
        "2-D fp32 row-major device tensor"
    return max(t.stride(0), t.shape[1]) if t.shape[0] > 1 else t.shape[1]


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           act: str = "none", gate: Optional[torch.Tensor] = None, rows_per_gate: int = 1,
           residual: Optional[torch.Tensor] = None, table: Optional[torch.Tensor] = None,
           partial: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(a @ w.T + bias) * gate[row // rows_per_gate] + residual + table[row % len(table)]  (one launch)."""
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k
    if out is None:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    g = CdxGemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=_p(bias), gate=_p(gate), residual=_p(residual),
                    table=_p(table), C=out.data_ptr(), M=m, N=n, K=k, lda=_rows(a), ldw=_rows(w), ldc=_rows(out),
                    ldg=_rows(gate) if gate is not None else 0, ldr=_rows(residual) if residual is not None else 0,
                    rows_per_gate=rows_per_gate, table_rows=table.shape[0] if table is not None else 0, act=ACT[act],
                    partial=_p(partial), partial_slices=(partial.numel() // max(m * n, 1)) if partial is not None else 0)
    if table is not None:
        assert table.shape[1] == n and table.is_contiguous()
    _check(_lib().cdx_gemm_f32(ctypes.byref(g), _stream_ptr(a.device)), "cdx_gemm_f32")
    return out


def conv1d(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], batch: int, l_in: int, stride: int = 1,
           pad: int = 0, out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: str = "none",
           l_out: Optional[int] = None, partial: Optional[torch.Tensor] = None):
    """Implicit-GEMM Conv1d on channel-last rows.  x: (batch*l_in, c_in); w_packed: (c_out, taps, c_in) = weight.permute(0,2,1);
    -> (batch*l_out, c_out), l_out = (l_in + 2 pad - taps) // stride + 1 unless given (`pad` is the LEFT padding; positions past
    either end read zeros, so an explicit l_out expresses asymmetric padding)."""
    n, taps, cin = w_packed.shape
    if l_out is None:
        l_out = (l_in + 2 * pad - taps) // stride + 1
    m = batch * l_out
    if out is None:
        out = torch.empty((m, n), device=x.device, dtype=torch.float32)
    w2 = w_packed.reshape(n, taps * cin)
    g = CdxGemmArgs(A=x.data_ptr(), W=w2.data_ptr(), bias=_p(bias), residual=_p(residual), C=out.data_ptr(), M=m, N=n,
                    K=taps * cin, lda=_rows(x), ldw=_rows(w2), ldc=_rows(out), ldr=_rows(residual) if residual is not None else 0,
                    rows_per_gate=1, act=ACT[act], conv_taps=taps, conv_cin=cin, conv_lin=l_in, conv_lout=l_out,
                    conv_stride=stride, conv_pad=pad, partial=_p(partial),
                    partial_slices=(partial.numel() // (m * n)) if partial is not None else 0)
    _check(_lib().cdx_gemm_f32(ctypes.byref(g), _stream_ptr(x.device)), "cdx_gemm_f32(conv)")
    return out


def pack_conv(weight: torch.Tensor) -> torch.Tensor:
    """nn.Conv1d weight (c_out, c_in, k) -> (c_out, k, c_in): K index = tap * c_in + c, contiguous in c."""
    return weight.detach().permute(0, 2, 1).contiguous()


def pack_conv_transpose_k4s2p1(weight: torch.Tensor):
    """nn.ConvTranspose1d(k=4, stride=2, pad=1) weight (c_in, c_out, 4) -> two (c_out, 2, c_in) stride-1 kernels, one per output
    parity:  out[2j]   = W[:, :, 3]^T x[j-1] + W[:, :, 1]^T x[j]     (taps at shifts -1, 0  -> pad 1)
             out[2j+1] = W[:, :, 2]^T x[j]   + W[:, :, 0]^T x[j+1]   (taps at shifts  0, +1 -> pad 0)"""
    w = weight.detach().permute(1, 2, 0)                      # (c_out, 4, c_in)
    # (stack of slices, not w[:, [3, 1]]: a list index becomes a host -> device copy of an index tensor, which a HIP-graph capture of
    #  the training step refuses)
    return torch.stack((w[:, 3], w[:, 1]), dim=1).contiguous(), torch.stack((w[:, 2], w[:, 0]), dim=1).contiguous()


def conv_transpose1d_k4s2p1(x: torch.Tensor, packed, bias, batch: int, l_in: int, out: Optional[torch.Tensor] = None):
    """ConvTranspose1d(k=4, s=2, p=1) as two implicit-GEMM convs writing the even / odd output rows (ldc = 2 N)."""
    even, odd = packed
    n = even.shape[0]
    if out is None:
        out = torch.empty((batch * l_in * 2, n), device=x.device, dtype=torch.float32)
    view = out.view(batch * l_in, 2 * n)                      # row (b, j) holds [out[2j] | out[2j+1]]
    conv1d(x, even, bias, batch, l_in, 1, 1, out=view[:, :n], l_out=l_in)
    conv1d(x, odd, bias, batch, l_in, 1, 0, out=view[:, n:], l_out=l_in)
    return out


def groupnorm(x: torch.Tensor, gamma, beta, batch: int, length: int, groups: int, act: str = "none", eps: float = 1e-5,
              fa=None, fb=None, fa_row: int = 0, fa_per_sample: bool = False, film_mode: int = 0, residual=None,
              out: Optional[torch.Tensor] = None):
    c = x.shape[1]
    if out is None:
        out = torch.empty_like(x)
    a = CdxGnArgs(x=x.data_ptr(), y=out.data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(), fa=_p(fa), fb=_p(fb),
                  residual=_p(residual), B=batch, L=length, C=c, G=groups, ldx=_rows(x), ldy=_rows(out),
                  ldr=_rows(residual) if residual is not None else 0, ldfa=_rows(fa) if fa is not None else 0,
                  ldfb=_rows(fb) if fb is not None else 0, fa_row=fa_row, fa_per_sample=int(fa_per_sample), film_mode=film_mode,
                  act=ACT[act], eps=eps)
    _check(_lib().cdx_groupnorm_f32(ctypes.byref(a), _stream_ptr(x.device)), "cdx_groupnorm_f32")
    return out


def groupnorm_backward(dy: torch.Tensor, x: torch.Tensor, gamma, beta, batch: int, length: int, groups: int,
                       act: str = "mish", eps: float = 1e-5, out: Optional[torch.Tensor] = None, param_grads: bool = False,
                       grads_out=None, possum_out: Optional[torch.Tensor] = None):
    """d loss / d x for y = act(groupnorm(x) * gamma + beta), given dy = d loss / d y (x is the saved forward input).
    `param_grads`: also (d loss / d gamma, d loss / d beta) -- per-sample partial sums out of the same launch, summed over the batch by
    cdx_colsum_f32 -> (dx, dgamma, dbeta).  `grads_out` = (tensor, tensor): the two sums are ADDED to these (the parameters'
    ``.grad``) by the backward kernel itself (float atomics; no staging, no column-sum launches) -> (dx, None, None).
    `possum_out` (batch, C) [a column block of a wider matrix is fine]: the launch also writes dy summed over each sample's positions."""
    if out is None:
        out = torch.empty_like(x)
    c = x.shape[1]
    if possum_out is not None:
        assert possum_out.shape == (batch, c) and possum_out.stride(1) == 1 and possum_out.dtype == torch.float32
    pg = pb = None
    if grads_out is not None:
        assert param_grads and all(t.is_contiguous() and t.numel() == c and t.dtype == torch.float32 for t in grads_out)
    elif param_grads:
        pg = torch.empty((2, batch, c), device=x.device, dtype=torch.float32)
        pb = pg[1]
    a = CdxGnArgs(x=x.data_ptr(), y=out.data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(), residual=dy.data_ptr(),
                  B=batch, L=length, C=c, G=groups, ldx=_rows(x), ldy=_rows(out), ldr=_rows(dy), act=ACT[act], eps=eps,
                  dgamma_part=None if pg is None else pg[0].data_ptr(), dbeta_part=None if pb is None else pb.data_ptr(),
                  dgamma_sum=None if grads_out is None else grads_out[0].data_ptr(), dbeta_sum=None if grads_out is None else grads_out[1].data_ptr(),
                  dy_possum=_p(possum_out), ld_possum=_rows(possum_out) if possum_out is not None else 0)
    _check(_lib().cdx_groupnorm_bwd_f32(ctypes.byref(a), _stream_ptr(x.device)), "cdx_groupnorm_bwd_f32")
    if not param_grads:
        return out
    # (the two (batch, C) partial blocks lie back to back: summed as 2 x batch rows into one zeroed (2, C) block by two launches that
    #  share the memset)
    if grads_out is not None:                             # (added by the kernel itself: float atomics onto the two (C) tensors)
        return out, None, None
    g = torch.zeros((2, c), device=x.device, dtype=torch.float32)
    colsum(pg[0], out=g[0])
    colsum(pg[1], out=g[1])
    return out, g[0], g[1]


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum over the rows of a (rows, cols) fp32 device matrix, ADDED to `out` (zeros when not given): one launch, float atomics."""
    r, c = x.shape
    if out is None:
        out = torch.zeros(c, device=x.device, dtype=torch.float32)
    _check(_lib().cdx_colsum_f32(x.data_ptr(), out.data_ptr(), r, c, _rows(x), _stream_ptr(x.device)), "cdx_colsum_f32")
    return out


def relayout_table(jobs, device):
    """Device tables for ``relayout``: `jobs` = [(src tensor, element offset of index (0, 0, 0), dst tensor (contiguous), (n0, n1, n2),
    (s0, s1, s2))] -> (jobs bytes on the device, chunk pairs on the device, n_chunks).  The tensors must stay alive and in place for as
    long as the table is used (their addresses are in it)."""
    import numpy as np
    arr = (CdxRelayoutJob * len(jobs))()
    chunks = []
    for j, (src, off, dst, n, st) in enumerate(jobs):
        assert src.dtype == dst.dtype == torch.float32 and dst.is_contiguous() and dst.numel() == n[0] * n[1] * n[2]
        arr[j] = CdxRelayoutJob(src=src.data_ptr() + 4 * off, dst=dst.data_ptr(), n0=n[0], n1=n[1], n2=n[2], s0=st[0], s1=st[1], s2=st[2])
        chunks += [(j, c) for c in range(-(-dst.numel() // RELAYOUT_CHUNK))]
    # pinned staging + asynchronous copies: the upload must not synchronise the stream (it may happen inside the capturability probe of
    # a training step, which treats every synchronising call as "this step cannot be captured"); the staging tensors stay alive with
    # the table
    host = [torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()), torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2)]
    if torch.device(device).type == "cuda":
        host = [h.pin_memory() for h in host]
    raw, ck = (h.to(device, non_blocking=True) for h in host)
    return raw, ck, len(chunks), host


def relayout(table):
    """ONE launch: every job of a ``relayout_table``."""
    raw, ck, n = table[:3]
    _check(_lib().cdx_relayout_f32(raw.data_ptr(), ck.data_ptr(), n, _stream_ptr(raw.device)), "cdx_relayout_f32")


def gather_windows(row0: torch.Tensor, fields, rows: int):
    """One launch: for every (src, steps) in `fields` -- src a (rows, width) fp32 device matrix -- the windows
    out[b] = src[row0[b] : row0[b] + steps] -> list of (batch, steps, width) tensors.  `row0`: int32 device vector; the caller
    guarantees row0 + steps <= rows (reference: D4RLMuJoCoDataset.__getitem__, dataset/d4rl_mujoco_dataset.py:138-151, per item on
    the host)."""
    assert row0.dtype == torch.int32 and row0.is_cuda and row0.dim() == 1 and row0.is_contiguous()
    assert 0 < len(fields) <= GATHER_MAX_FIELDS
    batch = row0.shape[0]
    a = CdxGatherArgs(row0=row0.data_ptr(), batch=batch, n_fields=len(fields), rows=rows)
    outs = []
    for i, (src, steps) in enumerate(fields):
        assert src.dtype == torch.float32 and src.is_cuda and src.dim() == 2 and src.is_contiguous() and src.shape[0] == rows
        out = torch.empty(batch, steps, src.shape[1], device=src.device, dtype=torch.float32)
        a.field[i] = CdxGatherField(src=src.data_ptr(), out=out.data_ptr(), width=src.shape[1], steps=steps)
        outs.append(out)
    _check(_lib().cdx_gather_windows_f32(ctypes.byref(a), _stream_ptr(row0.device)), "cdx_gather_windows_f32")
    return outs


def conv_wgrad(p: torch.Tensor, q: torch.Tensor, batch: int, l_p: int, l_q: int, taps: int, stride: int = 1, pad: int = 0,
               k_split: int = 0, bias_grad: bool = False, dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """dw[a][b][t] = sum_{n, m} p[(n, m)][a] * q[(n, m * stride + t - pad)][b] -> (ca, cb, taps), the layout of nn.Conv1d.weight with
    p = d loss / d y, q = x (and of nn.ConvTranspose1d.weight with p = x, q = d loss / d y, stride 2, pad 1).  One launch.
    `bias_grad` (p = d loss / d y): also the column sums of p out of the same launch -> (dw, db); both live in ONE zeroed buffer.
    `dw_out` (ca * cb * taps contiguous floats) [/ `db_out` (ca)]: ADD the sums to these tensors instead (a parameter's ``.grad``:
    the kernel accumulates with float atomics either way) -- nothing is allocated or zeroed."""
    ca, cb = p.shape[1], q.shape[1]
    assert p.shape[0] == batch * l_p and q.shape[0] == batch * l_q
    n_w = ca * cb * taps
    if dw_out is not None:
        assert dw_out.is_contiguous() and dw_out.numel() == n_w and dw_out.dtype == torch.float32
        assert not bias_grad or (db_out is not None and db_out.is_contiguous() and db_out.numel() == ca and db_out.dtype == torch.float32)
        dw, db = dw_out, (db_out if bias_grad else None)
    else:
        buf = torch.zeros(n_w + (ca if bias_grad else 0), device=p.device, dtype=torch.float32)
        dw = buf[:n_w].view(ca, cb, taps)
        db = buf[n_w:] if bias_grad else None
    a = CdxWgradArgs(p=p.data_ptr(), q=q.data_ptr(), dw=dw.data_ptr(), batch=batch, l_p=l_p, l_q=l_q, ca=ca, cb=cb, taps=taps,
                     stride=stride, pad=pad, ldp=_rows(p), ldq=_rows(q), k_split=k_split, db=_p(db))
    _check(_lib().cdx_conv_wgrad_f32(ctypes.byref(a), _stream_ptr(p.device)), "cdx_conv_wgrad_f32")
    return (dw, db) if bias_grad else dw


# workgroups a queued product is cut into (tiles x taps x row slices), at least WGRAD_MIN_CHUNKS 16-row chunks each (tuning hooks:
# CDX_WGRAD_JOB_WGS / CDX_WGRAD_BATCH_MIN_CHUNKS; sweep in profiles/r06_wgrad_batch_ab.txt)
# measured on MI355X (update() as a HIP graph, ms for configs 2 / 3 / 4): 384 workgroups x >= 8 chunks 3.35 / 9.48 / 2.33; 128 x 16: 3.12 / 9.29 /
# 2.40; 64 x 16: 3.07 / 9.48 / 2.49 -- convolutions (taps > 1) take 128, Linears (one tap, wide tiles: the transformers) 384
WGRAD_JOB_WGS = int(os.environ.get("CDX_WGRAD_JOB_WGS", "0"))
WGRAD_MIN_CHUNKS = int(os.environ.get("CDX_WGRAD_BATCH_MIN_CHUNKS", "16"))


def conv_wgrad_batch(jobs) -> int:
    """The weight-gradient products ``(p, q, batch, l_p, l_q, taps, stride, pad, dw_out, db_out | None)`` of many layers, ADDED into
    their ``dw_out`` / ``db_out`` tensors (parameters' ``.grad``) by ceil(n / 32) launches of ``cdx_conv_wgrad_batch_f32``.  Each
    product is cut into ~WGRAD_JOB_WGS workgroups: big layers get few, long row slices (every slice ends in one float atomic per
    output element), small layers many.  -> launches issued."""
    n_launch = 0
    for lo in range(0, len(jobs), WGRAD_BATCH):
        part = jobs[lo:lo + WGRAD_BATCH]
        b = CdxWgradBatch()
        b.n_jobs = len(part)
        start = 0
        for j, (p, q, batch, l_p, l_q, taps, stride, pad, dw, db) in enumerate(part):
            ca, cb = p.shape[1], q.shape[1]
            assert p.shape[0] == batch * l_p and q.shape[0] == batch * l_q and dw.is_contiguous() and dw.numel() == ca * cb * taps
            tiles = -(-ca // 64) * -(-cb // 64) * taps
            chunks = -(-(batch * l_p) // 16)
            target = WGRAD_JOB_WGS or (128 if taps > 1 else 384)
            ks = max(1, min(chunks // WGRAD_MIN_CHUNKS, -(-target // tiles)))
            b.wg_start[j] = start
            start += tiles * ks
            b.job[j] = CdxWgradArgs(p=p.data_ptr(), q=q.data_ptr(), dw=dw.data_ptr(), batch=batch, l_p=l_p, l_q=l_q, ca=ca, cb=cb, taps=taps,
                                    stride=stride, pad=pad, ldp=_rows(p), ldq=_rows(q), k_split=ks, db=_p(db))
        b.wg_start[len(part)] = start
        _check(_lib().cdx_conv_wgrad_batch_f32(ctypes.byref(b), _stream_ptr(part[0][0].device)), "cdx_conv_wgrad_batch_f32")
        n_launch += 1
    return n_launch


def layernorm(x: torch.Tensor, out: Optional[torch.Tensor] = None, gamma=None, beta=None, scale=None, shift=None,
              rows_per_mod: int = 1, eps: float = 1e-5) -> torch.Tensor:
    m, c = x.shape
    if out is None:
        out = torch.empty_like(x)
    a = CdxLnArgs(x=x.data_ptr(), y=out.data_ptr(), gamma=_p(gamma), beta=_p(beta), scale=_p(scale), shift=_p(shift),
                  M=m, C=c, ldx=_rows(x), ldy=_rows(out), ldmod=_rows(scale) if scale is not None else 0,
                  rows_per_mod=rows_per_mod, eps=eps)
    _check(_lib().cdx_layernorm_f32(ctypes.byref(a), _stream_ptr(x.device)), "cdx_layernorm_f32")
    return out


def layernorm_backward(dy: torch.Tensor, x: torch.Tensor, gamma=None, scale=None, rows_per_mod: int = 1, eps: float = 1e-5,
                       want_dyxhat: bool = False):
    """Backward of ``layernorm``: dx, and (optionally) dy * xhat per element -- what the gain / scale gradient sums (cdx.h)."""
    m, c = x.shape
    assert dy.shape == x.shape and dy.dtype == x.dtype == torch.float32
    dx = torch.empty((m, c), device=x.device, dtype=torch.float32)
    dyx = torch.empty((m, c), device=x.device, dtype=torch.float32) if want_dyxhat else None
    a = CdxLnBwdArgs(x=x.data_ptr(), dy=dy.data_ptr(), dx=dx.data_ptr(), dyxhat=_p(dyx), gamma=_p(gamma), scale=_p(scale), M=m, C=c,
                     ldx=_rows(x), lddy=_rows(dy), lddx=c, ldmod=_rows(scale) if scale is not None else 0, rows_per_mod=rows_per_mod, eps=eps)
    _check(_lib().cdx_layernorm_bwd_f32(ctypes.byref(a), _stream_ptr(x.device)), "cdx_layernorm_bwd_f32")
    return (dx, dyx) if want_dyxhat else dx


def attention_backward(qkv: torch.Tensor, dout: torch.Tensor, batch: int, tokens: int, n_heads: int) -> torch.Tensor:
    """d qkv of ``attention`` without a mask (probabilities recomputed from qkv)."""
    dm = qkv.shape[1] // 3
    assert qkv.is_contiguous() and dout.is_contiguous() and qkv.shape[0] == batch * tokens and dout.shape == (batch * tokens, dm)
    dqkv = torch.empty_like(qkv)
    dh = dm // n_heads
    a = CdxAttnBwdArgs(qkv=qkv.data_ptr(), dout=dout.data_ptr(), dqkv=dqkv.data_ptr(), B=batch, T=tokens, n_heads=n_heads, head_dim=dh,
                       scale=float(dh) ** -0.5)
    _check(_lib().cdx_attention_bwd_f32(ctypes.byref(a), _stream_ptr(qkv.device)), "cdx_attention_bwd_f32")
    return dqkv


def mha_train(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, n_heads: int, mask: Optional[torch.Tensor] = None,
              keep: Optional[torch.Tensor] = None, dout: Optional[torch.Tensor] = None, grads=None):
    """The attention core of nn.MultiheadAttention in training (``cdx_mha_train_fwd_f32`` / ``_bwd_f32``): q (batch * Tq, dm), k / v
    (batch * Tk, dm) -- column blocks of packed projections welcome --, additive `mask` (Tq, Tk), dropout `keep` (batch, n_heads, Tq,
    Tk) of 0 | 1 / (1 - p).  Without `dout`: the output (batch * Tq, dm); with it: (dq, dk, dv) -- written into `grads` (three
    2-D views, e.g. the column blocks of one packed gradient) when given, else three fresh tensors."""
    dm = q.shape[1]
    tq, tk, dh = q.shape[0] // batch, k.shape[0] // batch, dm // n_heads
    assert q.shape[0] == batch * tq and k.shape == v.shape == (batch * tk, dm) and dh * n_heads == dm
    if mask is not None:
        assert mask.shape == (tq, tk) and mask.is_contiguous() and mask.dtype == torch.float32
    if keep is not None:
        assert keep.shape == (batch, n_heads, tq, tk) and keep.is_contiguous() and keep.dtype == torch.float32
    a = CdxMhaTrainArgs(q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), mask=_p(mask), keep=_p(keep), B=batch, Tq=tq, Tk=tk, n_heads=n_heads,
                        head_dim=dh, ldq=_rows(q), ldk=_rows(k), ldv=_rows(v), ldo=dm, scale=float(dh) ** -0.5)
    if dout is None:
        out = torch.empty((batch * tq, dm), device=q.device, dtype=torch.float32)
        a.out = out.data_ptr()
        _check(_lib().cdx_mha_train_fwd_f32(ctypes.byref(a), _stream_ptr(q.device)), "cdx_mha_train_fwd_f32")
        return out
    assert dout.is_contiguous() and dout.shape == (batch * tq, dm)
    dq, dk, dv = grads if grads is not None else (torch.empty_like(dout), k.new_empty((batch * tk, dm)), k.new_empty((batch * tk, dm)))
    assert dq.shape == (batch * tq, dm) and dk.shape == dv.shape == (batch * tk, dm)
    a.dout, a.dq, a.dk, a.dv = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.lddq, a.lddk, a.lddv = _rows(dq), _rows(dk), _rows(dv)
    _check(_lib().cdx_mha_train_bwd_f32(ctypes.byref(a), _stream_ptr(q.device)), "cdx_mha_train_bwd_f32")
    return dq, dk, dv


def attention(qkv: torch.Tensor, batch: int, tokens: int, n_heads: int, out: Optional[torch.Tensor] = None,
              mask: Optional[torch.Tensor] = None):
    dm = qkv.shape[1] // 3
    assert qkv.is_contiguous() and qkv.shape[0] == batch * tokens
    if out is None:
        out = torch.empty((batch * tokens, dm), device=qkv.device, dtype=torch.float32)
    dh = dm // n_heads
    a = CdxAttnArgs(qkv=qkv.data_ptr(), out=out.data_ptr(), B=batch, T=tokens, n_heads=n_heads, head_dim=dh,
                    scale=float(dh) ** -0.5, mask=_p(mask))
    if mask is not None:
        assert mask.shape == (tokens, tokens) and mask.is_contiguous() and mask.dtype == torch.float32
    _check(_lib().cdx_attention_f32(ctypes.byref(a), _stream_ptr(qkv.device)), "cdx_attention_f32")
    return out


def cross_attention(q: torch.Tensor, kv_shared: torch.Tensor, kv_rows: Optional[torch.Tensor], batch: int, tokens: int,
                    n_obs: int, n_heads: int, shared_row: int = 0, shared_per_sample: bool = False,
                    mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q [k_shared | k_rows]^T / sqrt(dh) + mask) [v_shared | v_rows] per (sample, head); kv = [k | v] columns."""
    dm = q.shape[1]
    dh = dm // n_heads
    if out is None:
        out = torch.empty_like(q)
    a = CdxXattnArgs(q=q.data_ptr(), kv_shared=kv_shared.data_ptr(), kv_rows=_p(kv_rows), mask=_p(mask), out=out.data_ptr(),
                     B=batch, T=tokens, n_obs=n_obs, n_heads=n_heads, head_dim=dh, shared_row=shared_row,
                     shared_per_sample=int(shared_per_sample), scale=float(dh) ** -0.5)
    _check(_lib().cdx_cross_attention_f32(ctypes.byref(a), _stream_ptr(q.device)), "cdx_cross_attention_f32")
    return out


def activation(x: torch.Tensor, act: str, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert x.is_contiguous() and x.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    _check(_lib().cdx_act_f32(x.data_ptr(), out.data_ptr(), x.numel(), ACT[act], _stream_ptr(x.device)), "cdx_act_f32")
    return out


def activation_backward(pre: torch.Tensor, g: torch.Tensor, act: str, param: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g * act'(pre) elementwise (one launch): the activation factor of an explicit MLP backward pass."""
    assert pre.is_contiguous() and g.is_contiguous() and pre.shape == g.shape and pre.dtype == g.dtype == torch.float32
    if out is None:
        out = torch.empty_like(pre)
    _check(_lib().cdx_act_bwd_f32(pre.data_ptr(), g.data_ptr(), out.data_ptr(), pre.numel(), ACT[act], float(param),
                                  _stream_ptr(pre.device)), "cdx_act_bwd_f32")
    return out
