"""ctypes front end of the big-batch building blocks of libcdx.so (include/cdx.h: cdx_gemm_f32, cdx_layernorm_f32,
cdx_attention_f32, cdx_act_f32).  Every tensor crosses as a raw device pointer; outputs are caller-allocated.
These are used when M = batch x tokens >> 256 (DiT1d, wide MLPs), where the layers are classic GEMMs."""
import ctypes
from typing import Optional

import torch

from . import program as P
from .runtime import _check, _stream_ptr, load_library

ACT = {"none": P.ACT_NONE, "mish": P.ACT_MISH, "gelu": P.ACT_GELU_ERF, "leaky": P.ACT_LEAKY, "silu": P.ACT_SILU,
       "relu": P.ACT_RELU, "gelu_tanh": P.ACT_GELU_TANH}


class CdxGemmArgs(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("gate", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("table", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("lda", ctypes.c_int32),
                ("ldw", ctypes.c_int32), ("ldc", ctypes.c_int32), ("ldg", ctypes.c_int32), ("ldr", ctypes.c_int32),
                ("rows_per_gate", ctypes.c_int32), ("table_rows", ctypes.c_int32), ("act", ctypes.c_int32)]


class CdxLnArgs(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("C", ctypes.c_int32), ("ldx", ctypes.c_int32), ("ldy", ctypes.c_int32),
                ("ldmod", ctypes.c_int32), ("rows_per_mod", ctypes.c_int32), ("eps", ctypes.c_float),
                ("x_rows", ctypes.c_int32)]


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
        lib.cdx_cross_attention_f32.argtypes = [ctypes.POINTER(CdxXattnArgs), ctypes.c_void_p]
        lib.cdx_cross_attention_f32.restype = ctypes.c_int
        lib.cdx_act_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
        for f in (lib.cdx_gemm_f32, lib.cdx_layernorm_f32, lib.cdx_attention_f32, lib.cdx_act_f32):
            f.restype = ctypes.c_int
        _declared = True
    return lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _rows(t: torch.Tensor):
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, "2-D fp32 row-major device tensor"
    return t.stride(0)


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           act: str = "none", gate: Optional[torch.Tensor] = None, rows_per_gate: int = 1,
           residual: Optional[torch.Tensor] = None, table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(a @ w.T + bias) * gate[row // rows_per_gate] + residual + table[row % len(table)]  (one launch)."""
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k
    if out is None:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    g = CdxGemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=_p(bias), gate=_p(gate), residual=_p(residual),
                    table=_p(table), C=out.data_ptr(), M=m, N=n, K=k, lda=_rows(a), ldw=_rows(w), ldc=_rows(out),
                    ldg=_rows(gate) if gate is not None else 0, ldr=_rows(residual) if residual is not None else 0,
                    rows_per_gate=rows_per_gate, table_rows=table.shape[0] if table is not None else 0, act=ACT[act])
    if table is not None:
        assert table.shape[1] == n and table.is_contiguous()
    _check(_lib().cdx_gemm_f32(ctypes.byref(g), _stream_ptr(a.device)), "cdx_gemm_f32")
    return out


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
