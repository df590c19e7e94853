"""Training step of the U-Net denoisers on the library's kernels: the forward and backward of
``DiffusionModel.update()`` (reference cleandiffuser/diffusion/diffusionsde.py:94-141 ``loss`` / ``update``, basic.py:66,83-86).

The reference trains through PyTorch autograd over ``nn.Conv1d`` / ``F.group_norm`` / ``nn.Mish``: ~170 ATen launches forward and as
many again backward, the convolutions on MIOpen kernels that are tuned for images, not for (256 x 32 x 32)-sized temporal layers.  Here
the network keeps its ``nn.Module`` parameters (checkpoints, optimiser, EMA unchanged) and autograd keeps the GRAPH -- but every node that
touches a convolution or a GroupNorm is a ``torch.autograd.Function`` whose forward and backward are library launches on channel-last
``(batch * positions, channels)`` rows (the layout the sampling executors use; the (b, H, D) trajectory tensor IS that layout, so
nothing is transposed on the way in or out):

* ``Conv1d`` / ``ConvTranspose1d(4, 2, 1)``: forward = ``cdx_gemm_f32`` implicit-GEMM conv; backward-data = the same kernel with flipped,
  transposed weights (a stride-2 conv: two parity convs; a transposed conv: a stride-2 conv); weight gradient = ``cdx_conv_wgrad_f32``
  (a TN GEMM over the batch x position rows); bias gradient = ``cdx_colsum_f32``.
* ``GroupNorm -> Mish``: ``cdx_groupnorm_f32`` / ``cdx_groupnorm_bwd_f32`` (which also emits the per-sample partials of the gain / shift
  gradients; summed by ``cdx_colsum_f32``).
* what is left to ATen are the broadcast adds (FiLM vector, residual), the channel concat of the skip connections and the embedding MLP
  (a few (batch, 32..128) Linears): no convolution, no group_norm.

``CDX_TRAIN_NATIVE=0`` keeps the reference's autograd path (A/B runs, the fixtures' twin).  Parity: tests/test_gpu_parity.py --
gradients of every parameter against ``torch.autograd`` of the module, and the ``train_*`` fixtures of the real reference.
"""
import contextlib
import functools
import os
import threading
from typing import Optional

import torch
import torch.nn as nn

from . import blocks


def enabled() -> bool:
    return os.environ.get("CDX_TRAIN_NATIVE", "1") != "0"


def supports(net, x: torch.Tensor, condition=None) -> bool:
    """JannerUNet1d (GroupNorm, no attention) with fp32 parameters on a ROCm device, called with autograd on."""
    from .consts import supports_janner
    from .runtime import _is_janner
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and _is_janner(net)):
        return False
    if supports_janner(net) is not None or net.kernel_size > 5:
        return False
    if not _groupnorms_ok(net) or not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


def _groupnorms_ok(net) -> bool:
    """cdx_groupnorm_bwd_f32 reduces the gain / shift gradients per channel of a group of C / G channels: a power of two, at most 256
    (model_dim 48 -- groups of 6 / 12 -- does not fit): such nets keep the reference's ATen autograd path instead of failing inside
    backward()."""
    for m in net.modules():
        if isinstance(m, nn.GroupNorm) or type(m).__name__ == "GroupNorm1d":
            c, g = m.weight.shape[0], max(int(m.num_groups), 1)
            cg = c // g
            if c % g or cg > 256 or cg & (cg - 1):
                return False
    return True


def _wants_grad(net, *tensors) -> bool:
    """Does anything in this forward take part in autograd?  A frozen net (model_ema) called without no_grad() on inputs that need no
    gradient belongs to the one-launch fused forward, not to the per-layer training graph."""
    if any(getattr(t, "requires_grad", False) for t in tensors):
        return True
    return any(p.requires_grad for p in net.parameters())


# --------------------------------------------------------------------------------------------------------------------- #
# Parameter gradients of update(): accumulated where they live                                                                 #
# --------------------------------------------------------------------------------------------------------------------- #
# The weight / bias / gain sums of a layer come out of kernels that ADD into their output with float atomics (cdx_conv_wgrad_f32,
# cdx_colsum_f32).  Handing such a sum to autograd costs three more launches per parameter pair: the zero-fill of a fresh buffer and one
# `grad += new` each (AccumulateGrad) -- ~250 launches of the ~970 of a config-2 step (profiles/r05_update_census.txt).  Inside update()'s own
# backward pass (DiffusionModel._loss_backward, and the HIP graph captured from it) the nodes therefore add straight into `p.grad` and
# report "no gradient" to autograd.  Anywhere else -- torch.autograd.grad(), a user's own backward() over loss(), parameters with
# hooks, views of parameters -- autograd's own accumulation runs as before.
_in_place_depth = 0          # (module-wide, not thread-local: autograd runs the nodes of a device on its own worker thread)
_in_place_scopes: list = []  # one entry per open grads_in_place(): the ids of the parameters it covers, or None = every parameter


@contextlib.contextmanager
def grads_in_place(params=None):
    """`params`: the parameters of the agent whose update() this is -- only THEY take gradients in place while the scope is
    open (the flag is process-wide because autograd's worker threads run the nodes; another thread's backward over another net keeps
    autograd's own accumulation).  None: every parameter (tests, a caller that owns the process)."""
    global _in_place_depth
    scope = None if params is None else {id(p) for p in params}
    _in_place_depth += 1
    _in_place_scopes.append(scope)
    try:
        yield
    finally:
        _in_place_depth -= 1
        _in_place_scopes.remove(scope)


def _reducer_may_listen() -> bool:
    """A process group with more than one rank is up: a DistributedDataParallel / FSDP reducer may hang on the parameters' gradient
    accumulators (C++ post hooks, invisible from Python), and those fire only when autograd itself accumulates -- no in-place sums."""
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:        # noqa: BLE001
        return False


def _grad_slot(p) -> Optional[torch.Tensor]:
    """``p.grad`` as the tensor a kernel may add p's gradient into (created zeroed if missing), or None: autograd accumulates."""
    if _in_place_depth <= 0 or os.environ.get("CDX_TRAIN_INPLACE_GRADS", "1") == "0":
        return None
    if not any(sc is None or id(p) in sc for sc in _in_place_scopes) or _reducer_may_listen():
        return None
    if not isinstance(p, nn.Parameter) or not p.is_leaf or not p.requires_grad or p.dtype != torch.float32 or p._backward_hooks or \
            getattr(p, "_post_accumulate_grad_hooks", None):
        return None
    g = p.grad
    if g is None:
        g = p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    if g.dtype != torch.float32 or g.shape != p.shape or g.device != p.device or not g.is_contiguous() or g.requires_grad:
        return None
    return g


def _written(*grads):
    """The kernels wrote these ``.grad`` tensors behind autograd's back: move their version counters (FusedAdamW tells a gradient
    somebody wrote from one it zeroed itself by them)."""
    gs = [g for g in grads if g is not None]
    if gs:
        torch.autograd.graph.increment_version(gs)


# The weight-gradient products of a step are independent of everything downstream of them (they only ADD into ``.grad``): inside
# update()'s backward they are QUEUED by the nodes and issued when the backward pass ends, all layers in ceil(n / 32) launches of
# cdx_conv_wgrad_batch_f32 -- round 5 issued 65 launches of 4-20 us each, 38 % of a config-2 step's device time at 3.8 % of the
# matrix pipe's peak, none of them able to fill the chip.  CDX_TRAIN_WGRAD_BATCH=0: one launch per layer as before.
_wgrad_queues: dict = {}     # autograd graph-task id -> the products queued by that backward pass


def _queue_wgrad(job) -> bool:
    if _in_place_depth <= 0 or os.environ.get("CDX_TRAIN_WGRAD_BATCH", "1") == "0":
        return False
    task = torch._C._current_graph_task_id()
    if task < 0:
        return False                                       # not inside a backward pass (a node called by hand): launch now
    q = _wgrad_queues.get(task)
    if q is None:
        # a new backward pass (a re-entrant one inside another keeps its own list).  Lists that an EARLIER pass left behind -- it raised
        # before its end-of-pass callback ran -- belong to no step any more: drop them
        for old in [t for t in _wgrad_queues if t < task - 4]:
            del _wgrad_queues[old]
        q = []

        def flush(task=task):
            jobs = _wgrad_queues.pop(task, None)
            if jobs:
                blocks.conv_wgrad_batch(jobs)
        try:
            # (runs when this backward pass ends, on the stream it was called on -- what DDP's reducer relies on too)
            torch.autograd.Variable._execution_engine.queue_callback(flush)
        except RuntimeError:
            return False
        _wgrad_queues[task] = q
    q.append(job)
    return True


def _weight_grads(p_rows, q_rows, batch, l_p, l_q, taps, stride, pad, w_param, b_param, want_dw: bool, want_db: bool, bias_rows=None):
    """(dw, db) of a Conv1d / ConvTranspose1d / Linear for autograd -- each None when not wanted OR when it was added straight into
    the parameter's ``.grad`` (grads_in_place).  `bias_rows`: the matrix whose column sums are db when it is not `p_rows` (transposed
    convolution: p = x)."""
    dw = db = None
    fused_db = want_db and bias_rows is None              # db out of the weight-gradient launch itself
    if want_dw:
        sw, sb = _grad_slot(w_param), (_grad_slot(b_param) if fused_db else None)
        if sw is not None and (not fused_db or sb is not None):
            if not _queue_wgrad((p_rows, q_rows, batch, l_p, l_q, taps, stride, pad, sw, sb)):
                blocks.conv_wgrad(p_rows, q_rows, batch, l_p, l_q, taps, stride, pad, bias_grad=fused_db, dw_out=sw, db_out=sb)
            _written(sw, sb)
        else:
            dw = blocks.conv_wgrad(p_rows, q_rows, batch, l_p, l_q, taps, stride, pad, bias_grad=fused_db)
            if fused_db:
                dw, db = dw
            dw = dw.view(w_param.shape)
    if want_db and not (want_dw and fused_db):
        rows = p_rows if bias_rows is None else bias_rows
        sb = _grad_slot(b_param)
        if sb is not None:
            blocks.colsum(rows, out=sb)
            _written(sb)
        else:
            db = blocks.colsum(rows)
    return dw, db


# --------------------------------------------------------------------------------------------------------------------- #
# Weight layouts of a training step: one launch per forward pass                                                               #
# --------------------------------------------------------------------------------------------------------------------- #
# Every convolution / linear weight is needed in other layouts: taps-major for the implicit GEMM, flipped + transposed for the
# backward-data convolution, transposed for a Linear's input gradient.  As ATen permute / flip / stack / contiguous calls inside the
# nodes these were ~130 launches of a config-2 step (profiles/r05_update_census.txt).  Each is a 3-D strided gather of the weight, so a
# denoiser's whole set is ONE ``cdx_relayout_f32`` launch at the start of its forward pass (and nothing at all while the parameters
# have not changed): ``_WeightPacks`` keeps the job table.  A layout is registered the first time a node asks for it (that step still
# builds it with ATen); from the next parameter change on it is refreshed by the one launch.  CDX_TRAIN_PACKS=0: ATen calls as before.
def _geom(kind: str, w: torch.Tensor):
    """(dst shape, (n0, n1, n2), (s0, s1, s2), element offset of index (0, 0, 0)) of layout `kind` of weight `w` (its own strides:
    contiguous slices of a packed parameter are welcome)."""
    if kind == "linear_t":                                 # (A, B) -> (B, A)
        (a, b), (sa, sb) = w.shape, w.stride()
        return (b, a), (b, a, 1), (sb, sa, 0), 0
    (a, b, k), (sa, sb, sk) = w.shape, w.stride()
    if kind in ("conv", "convt_bwd"):                      # (A, B, K) -> (A, K, B)
        return (a, k, b), (a, k, b), (sa, sk, sb), 0
    if kind == "conv_bwd":                                 # -> (B, K, A), taps reversed
        return (b, k, a), (b, k, a), (sb, -sk, sa), (k - 1) * sk
    if kind == "conv_s2_mid":                              # -> (B, 1, A): tap 1
        return (b, 1, a), (b, 1, a), (sb, 0, sa), sk
    if kind in ("conv_s2_outer", "convt_odd"):             # -> (B, 2, A): taps 2, 0
        return (b, 2, a), (b, 2, a), (sb, -2 * sk, sa), 2 * sk
    if kind == "convt_even":                               # -> (B, 2, A): taps 3, 1
        return (b, 2, a), (b, 2, a), (sb, -2 * sk, sa), 3 * sk
    raise KeyError(kind)


def _aten_pack(kind: str, w: torch.Tensor) -> torch.Tensor:
    """The same layouts as ATen calls (what a node runs for a layout that is not registered yet, and what the tests compare with)."""
    w = w.detach()
    if kind == "linear_t":
        return w.t().contiguous()
    if kind in ("conv", "convt_bwd"):
        return w.permute(0, 2, 1).contiguous()
    if kind == "conv_bwd":
        return w.flip(2).permute(1, 2, 0).contiguous()
    v = w.permute(1, 2, 0)                                 # (B, K, A)
    if kind == "conv_s2_mid":
        return v[:, 1:2].contiguous()
    # (stack of slices, not v[:, [2, 0]]: a list index becomes a host -> device copy of an index tensor, which a HIP-graph capture of
    #  the training step refuses)
    if kind in ("conv_s2_outer", "convt_odd"):
        return torch.stack((v[:, 2], v[:, 0]), dim=1).contiguous()
    if kind == "convt_even":
        return torch.stack((v[:, 3], v[:, 1]), dim=1).contiguous()
    raise KeyError(kind)


class _WeightPacks:
    """The registered layouts of one module's weights (see above).  ``refresh`` at the start of a forward pass; nodes call ``get``."""

    def __init__(self):
        self.entries = {}          # (weight address, weight shape, kind) -> [dst, job, signature it is fresh for]
        self.table = None          # device tables of every entry registered when it was built
        self.in_table = 0
        self.sig = None
        self._listed, self._retired = [], []

    # The registry names ADDRESSES (of this module's parameters and of its own buffers): a copy of the module (``deepcopy(model)`` for the
    # EMA twin, a pickled module) starts with an empty one.
    def __deepcopy__(self, memo):
        return _WeightPacks()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def refresh(self, params):
        sig = tuple((p.data_ptr(), p._version) for p in params)
        dirty = self.in_table != len(self.entries)
        if torch.cuda.is_current_stream_capturing():
            # A capture records launches, it does not run them, and it cannot upload a table.  With a complete table the one launch
            # is recorded (a replay then refreshes every layout from the weights of that moment); without one NOTHING counts as fresh
            # in this pass, so every node records its own ATen calls instead.
            if dirty or not self.entries:
                self.sig = object()
                return
            self.sig = sig
            blocks.relayout(self.table)
        else:
            if self.sig is not None and not isinstance(self.sig, tuple):
                self.sig = None
            if self.sig is not None and [a for a, _ in sig] != [a for a, _ in self.sig]:
                # parameters moved: every address in the table is stale (the old homes stay alive: a captured graph may name them)
                self._retired.append((self.entries, self.table))
                self.entries, self.table, self.in_table, self._listed, dirty = {}, None, 0, [], False
            if dirty:                                      # layouts registered since the table was built (the table of a captured
                self._retired.append(self.table)           # graph stays alive)
                self._listed = list(self.entries.values())
                self.table = blocks.relayout_table([e[1] for e in self._listed], self._listed[0][0].device)
                self.in_table = len(self._listed)
            if sig == self.sig:
                return
            self.sig = sig
            if self.table is None:
                return
            blocks.relayout(self.table)
        for e in self._listed:
            e[2] = sig

    def get(self, kind: str, w: torch.Tensor) -> torch.Tensor:
        key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), kind)
        e = self.entries.get(key)
        if e is not None and e[2] == self.sig:
            return e[0]
        t = _aten_pack(kind, w)
        if t.untyped_storage().data_ptr() == w.untyped_storage().data_ptr():
            return t                                       # (a 1-tap / 1-column weight IS its own layout: a view, no launch, nothing to keep)
        if e is None and not torch.cuda.is_current_stream_capturing():
            # adopt this tensor as the layout's home: from the next parameter change on it is rewritten by the one launch
            shape, n, st, off = _geom(kind, w)
            assert tuple(t.shape) == shape
            self.entries[key] = [t, (w.detach(), off, t, n, st), self.sig]
        elif e is not None and not torch.cuda.is_current_stream_capturing():
            e[0].copy_(t)                                  # (registered but stale and not refreshed: keep its home current)
            e[2] = self.sig
            return e[0]
        return t


_tls = threading.local()     # .packs: the weight-layout registry of the forward pass THIS thread is in (nodes keep it on ctx for their backward)


def _current_packs() -> Optional[_WeightPacks]:
    return getattr(_tls, "packs", None)


@contextlib.contextmanager
def _weight_packs(net):
    """The forward pass of `net` on the library's nodes: its weight layouts are current inside (one launch, if anything changed)."""
    prev = _current_packs()
    packs = None
    if os.environ.get("CDX_TRAIN_PACKS", "1") != "0":
        packs = net.__dict__.get("_cdx_weight_packs")
        if packs is None:
            packs = net.__dict__["_cdx_weight_packs"] = _WeightPacks()
        packs.refresh(list(net.parameters()))
    _tls.packs = packs
    try:
        yield
    finally:
        _tls.packs = prev


def _pack(kind: str, w: torch.Tensor, packs: Optional[_WeightPacks]) -> torch.Tensor:
    return _aten_pack(kind, w) if packs is None else packs.get(kind, w)


def _with_weight_packs(forward):
    @functools.wraps(forward)
    def run(net, *args, **kwargs):
        with _weight_packs(net):
            return forward(net, *args, **kwargs)
    return run


def _splitk_scratch(rows: int, n: int, k: int, device) -> Optional[torch.Tensor]:
    """Scratch for a deterministic split-K launch of ``cdx_gemm_f32`` (cdx.h: partial) when the product has too few output tiles to
    fill the chip and a long K -- the training batches (256 rows x 1024..4096 columns over K = 1024..5120 for configs 3 / 5): without it
    a 64 x 64 tile walks the whole K alone on a quarter of the CUs.  None: the launch would not split anyway.  CDX_TRAIN_SPLITK=0: off."""
    if os.environ.get("CDX_TRAIN_SPLITK", "1") == "0":
        return None
    tiles_big = -(-rows // 128) * -(-n // 128)
    slices = min(16, k // int(os.environ.get("CDX_TRAIN_SPLITK_MINK", "256")))
    if tiles_big >= 192 or slices < 2 or rows * n * slices > (1 << 26):
        return None
    return torch.empty(slices * rows * n, device=device, dtype=torch.float32)


def _linear(a, w, bias=None):
    return blocks.linear(a, w, bias, partial=_splitk_scratch(a.shape[0], w.shape[0], w.shape[1], a.device))


# --------------------------------------------------------------------------------------------------------------------- #
class _Conv(torch.autograd.Function):
    """nn.Conv1d (stride 1 'same', or k = 3 / stride 2 / pad 1) on channel-last rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, batch, l_in, stride, pad):
        x = x.contiguous()
        ctx.packs = _current_packs()
        y = blocks.conv1d(x, _pack("conv", weight, ctx.packs), bias, batch, l_in, stride, pad,
                          partial=_splitk_scratch(x.shape[0] // stride, weight.shape[0], weight.shape[1] * weight.shape[2], x.device))
        ctx.save_for_backward(x, weight)
        ctx.geom = (batch, l_in, stride, pad, bias is not None)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        batch, l_in, stride, pad, has_bias = ctx.geom
        c_out, c_in, k = weight.shape
        dy = dy.contiguous()
        l_out = dy.shape[0] // batch
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if stride == 1:
                # dx[l] = sum_t W[:, :, t]^T dy[l + pad - t]: a conv of dy with the flipped, transposed kernel, left padding k - 1 - pad
                wt = _pack("conv_bwd", weight, ctx.packs)                                  # (c_in, k, c_out), taps reversed
                dx = blocks.conv1d(dy, wt, None, batch, l_out, 1, k - 1 - pad, l_out=l_in,
                                   partial=_splitk_scratch(batch * l_in, c_in, c_out * k, dy.device))
            else:
                assert (k, stride, pad) == (3, 2, 1) and l_in == 2 * l_out
                # y[m] = sum_t W_t x[2m - 1 + t]:  dx[2j] = W_1^T dy[j];  dx[2j + 1] = W_2^T dy[j] + W_0^T dy[j + 1]
                dx = torch.empty((batch * l_in, c_in), device=dy.device, dtype=torch.float32)
                view = dx.view(batch * l_out, 2 * c_in)                                    # row (b, j) = [dx[2j] | dx[2j + 1]]
                blocks.conv1d(dy, _pack("conv_s2_mid", weight, ctx.packs), None, batch, l_out, 1, 0, out=view[:, :c_in], l_out=l_out)
                blocks.conv1d(dy, _pack("conv_s2_outer", weight, ctx.packs), None, batch, l_out, 1, 0, out=view[:, c_in:], l_out=l_out)
        dw, db = _weight_grads(dy, x, batch, l_out, l_in, k, stride, pad, ctx.params[0], ctx.params[1], ctx.needs_input_grad[1],
                               has_bias and ctx.needs_input_grad[2])                        # (c_out, c_in, k), (c_out,)
        return dx, dw, db, None, None, None, None


class _ConvT(torch.autograd.Function):
    """nn.ConvTranspose1d(C, C', 4, 2, 1) on channel-last rows: L -> 2 L."""

    @staticmethod
    def forward(ctx, x, weight, bias, batch, l_in):
        x = x.contiguous()
        ctx.packs = _current_packs()
        y = blocks.conv_transpose1d_k4s2p1(x, (_pack("convt_even", weight, ctx.packs), _pack("convt_odd", weight, ctx.packs)), bias, batch, l_in)
        ctx.save_for_backward(x, weight)
        ctx.geom = (batch, l_in, bias is not None)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        batch, l_in, has_bias = ctx.geom
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # y[2m - 1 + t] += W[ci, :, t] x[m, ci]  =>  dx[m, ci] = sum_t W[ci, :, t] . dy[2m - 1 + t]: a stride-2 conv of dy
            wq = _pack("convt_bwd", weight, ctx.packs)                                     # (c_in, 4, c_out)
            dx = blocks.conv1d(dy, wq, None, batch, 2 * l_in, 2, 1)
        dw, db = _weight_grads(x, dy, batch, l_in, 2 * l_in, 4, 2, 1, ctx.params[0], ctx.params[1], ctx.needs_input_grad[1],
                               has_bias and ctx.needs_input_grad[2], bias_rows=dy)           # (c_in, c_out, 4), (c_out,)
        return dx, dw, db, None, None


class _GroupNormMish(torch.autograd.Function):
    """GroupNorm1d [-> Mish] on channel-last rows (reference utils/building_blocks.py:60-76 + nn.Mish); `act` "mish" (default) or "none"."""

    @staticmethod
    def forward(ctx, x, gamma, beta, batch, length, groups, eps, act="mish"):
        x = x.contiguous()
        y = blocks.groupnorm(x, gamma, beta, batch, length, groups, act=act, eps=eps)
        ctx.save_for_backward(x, gamma, beta)
        ctx.geom = (batch, length, groups, eps, act)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        batch, length, groups, eps, act = ctx.geom
        slots = (_grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])) if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] else (None, None)
        slots = slots if slots[0] is not None and slots[1] is not None else None
        dx, dg, db = blocks.groupnorm_backward(dy.contiguous(), x, gamma.detach(), beta.detach(), batch, length, groups, act=act, eps=eps,
                                               param_grads=True, grads_out=slots)
        if slots is not None:
            _written(*slots)
        return dx, dg, db, None, None, None, None, None


class _GroupNormMishAdd(torch.autograd.Function):
    """``Mish(GroupNorm1d(x)) [+ film[b]] [+ res]`` in ONE launch forward (cdx_groupnorm_f32: film_mode 2 and the residual operand) --
    the two adds of a ResidualBlock (reference jannerunet.py:66-69: ``conv1(x) + emb_mlp(emb)``, ``conv2(.) + residual_conv(x)``) were
    an ATen launch each, 32 of a config-2 step.  Each result element is still one fp32 add per term, as ATen's.  Backward: the
    additive terms pass dy through (`res`: as it is, no launch; `film`: summed over a sample's positions)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, film, res, batch, length, groups, eps):
        x = x.contiguous()
        y = blocks.groupnorm(x, gamma, beta, batch, length, groups, act="mish", eps=eps,
                             fb=None if film is None else (film if film.stride(1) == 1 else film.contiguous()),
                             film_mode=2 if film is not None else 0,
                             residual=None if res is None else res.contiguous())
        ctx.save_for_backward(x, gamma, beta)
        ctx.geom = (batch, length, groups, eps)
        ctx.params = (gamma, beta)
        ctx.has = (film is not None, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        batch, length, groups, eps = ctx.geom
        dy = dy.contiguous()
        slots = (_grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])) if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] else (None, None)
        slots = slots if slots[0] is not None and slots[1] is not None else None
        # (the FiLM vector's gradient -- dy summed over a sample's positions -- out of the same launch: cdx_gn_args.dy_possum, ABI 17;
        #  CDX_TRAIN_POSSUM=0: an ATen reduction per block as before)
        cgw = x.shape[1] // groups
        dfilm = None
        if ctx.has[0] and ctx.needs_input_grad[3] and cgw & (cgw - 1) == 0 and cgw <= 256 and os.environ.get("CDX_TRAIN_POSSUM", "1") != "0":
            dfilm = torch.empty((batch, x.shape[1]), device=x.device, dtype=torch.float32)
        dx, dg, db = blocks.groupnorm_backward(dy, x, gamma.detach(), beta.detach(), batch, length, groups, act="mish", eps=eps,
                                               param_grads=True, grads_out=slots, possum_out=dfilm)
        if slots is not None:
            _written(*slots)
        if dfilm is None and ctx.has[0] and ctx.needs_input_grad[3]:
            dfilm = dy.view(batch, length, -1).sum(1)
        dres = dy if ctx.has[1] and ctx.needs_input_grad[4] else None
        return dx, dg, db, dfilm, dres, None, None, None, None


class _ConvPair(torch.autograd.Function):
    """The two paths that leave a ResidualBlock's input (reference jannerunet.py:66-69): ``(conv1(x), residual_conv(x))`` -- the skip path
    a 1 x 1 Conv1d or the identity -- as ONE autograd node.  As two nodes autograd adds their input gradients with an ATen launch per
    block (16 of a config-2 step); here the second backward-data product takes the first path's gradient as the residual operand of
    its epilogue: dx = conv1^T(dy1) + [residual_conv^T](dres), one fp32 add per element as before."""

    @staticmethod
    def forward(ctx, x, w1, b1, wr, br, batch, l_in, pad):
        x = x.contiguous()
        ctx.packs = _current_packs()
        c_out, c_in, k = w1.shape
        y1 = blocks.conv1d(x, _pack("conv", w1, ctx.packs), b1, batch, l_in, 1, pad, partial=_splitk_scratch(x.shape[0], c_out, c_in * k, x.device))
        if wr is None:
            res = x.view_as(x)
        else:
            res = blocks.conv1d(x, _pack("conv", wr, ctx.packs), br, batch, l_in, 1, 0, partial=_splitk_scratch(x.shape[0], wr.shape[0], c_in, x.device))
        ctx.save_for_backward(x, w1, wr if wr is not None else x.new_empty(0))
        ctx.geom = (batch, l_in, pad, b1 is not None, wr is not None, br is not None)
        ctx.params = (w1, b1, wr, br)
        return y1, res

    @staticmethod
    def backward(ctx, dy1, dres):
        x, w1, wr = ctx.saved_tensors
        batch, l_in, pad, has_b1, has_wr, has_br = ctx.geom
        c_out, c_in, k = w1.shape
        dy1, dres = dy1.contiguous(), dres.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            d0 = dres
            if has_wr:
                d0 = blocks.conv1d(dres, _pack("conv_bwd", wr, ctx.packs), None, batch, l_in, 1, 0, l_out=l_in,
                                   partial=_splitk_scratch(batch * l_in, c_in, wr.shape[0], dres.device))
            dx = blocks.conv1d(dy1, _pack("conv_bwd", w1, ctx.packs), None, batch, l_in, 1, k - 1 - pad, l_out=l_in, residual=d0,
                               partial=_splitk_scratch(batch * l_in, c_in, c_out * k, dy1.device))
        dw1, db1 = _weight_grads(dy1, x, batch, l_in, l_in, k, 1, pad, ctx.params[0], ctx.params[1], ctx.needs_input_grad[1],
                                 has_b1 and ctx.needs_input_grad[2])
        dwr = dbr = None
        if has_wr:
            dwr, dbr = _weight_grads(dres, x, batch, l_in, l_in, 1, 1, 0, ctx.params[2], ctx.params[3], ctx.needs_input_grad[3],
                                     has_br and ctx.needs_input_grad[4])
        return dx, dw1, db1, dwr, dbr, None, None, None


class _LinearMany(torch.autograd.Function):
    """Several Linears of ONE input -- the emb_mlp Linears of every ResidualBlock of a U-Net, all fed Mish(emb) (reference
    jannerunet.py:60-66): ``(x W_0^T + b_0, x W_1^T + b_1, ...)`` as column blocks of one GEMM on the stacked weights.  Backward: the
    input's gradient is one GEMM on the stacked output gradients (sixteen GEMMs and fifteen ATen adds of a config-2 step as separate
    nodes); the weight gradients join the step's batch as before.  Arguments: x, then (weight, bias) per Linear."""

    @staticmethod
    def forward(ctx, x, *wb):
        x = x.contiguous()
        ws, bs = wb[0::2], wb[1::2]
        wcat = torch.cat([w.detach() for w in ws], 0)
        y = _linear(x, wcat, torch.cat([b.detach() for b in bs], 0))
        ctx.save_for_backward(x, wcat)
        ctx.params = wb
        outs, off = [], 0
        for w in ws:
            outs.append(y[:, off:off + w.shape[0]])
            off += w.shape[0]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x, wcat = ctx.saved_tensors
        ws, bs = ctx.params[0::2], ctx.params[1::2]
        dys = [d.contiguous() for d in dys]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _linear(torch.cat(dys, 1), wcat.t().contiguous())
        grads = []
        for i, (w, b, dy) in enumerate(zip(ws, bs, dys)):
            grads += _weight_grads(dy, x, x.shape[0], 1, 1, 1, 1, 0, w, b, ctx.needs_input_grad[1 + 2 * i], ctx.needs_input_grad[2 + 2 * i])
        return (dx, *grads)


def _films(blocks_list, memb):
    """{id(block): its FiLM vector (batch, c_out)} for the ResidualBlocks of a U-Net from one `_LinearMany` node (CDX_TRAIN_FILM_BATCH=0:
    empty -- every block runs its own Linear node, the round-5 graph)."""
    lins = [rb.emb_mlp[1] for rb in blocks_list]
    if os.environ.get("CDX_TRAIN_FILM_BATCH", "1") == "0" or len(lins) < 2 or any(lin.bias is None for lin in lins):
        return {}
    outs = _LinearMany.apply(memb, *[t for lin in lins for t in (lin.weight, lin.bias)])
    return {id(rb): o for rb, o in zip(blocks_list, outs)}


class _Act(torch.autograd.Function):
    """An elementwise activation on its own (``cdx_act_f32`` / ``cdx_act_bwd_f32``): behind a GroupNorm whose fused epilogue does not
    know it (the GELU of PearceMlp's FCBlock)."""

    @staticmethod
    def forward(ctx, z, act):
        z = z.contiguous()
        ctx.save_for_backward(z)
        ctx.act = act
        return blocks.activation(z, act)

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        return blocks.activation_backward(z, dy.contiguous(), ctx.act), None


# --------------------------------------------------------------------------------------------------------------------- #
def _conv(h, conv: nn.Conv1d, batch: int, length: int):
    return _Conv.apply(h, conv.weight, conv.bias, batch, length, conv.stride[0], conv.padding[0])


def _cna(h, seq: nn.Sequential, batch: int, length: int):
    """Conv1d -> GroupNorm1d -> Mish (a `_conv_norm_act` Sequential of nn_diffusion/jannerunet.py)."""
    conv, gn = seq[0], seq[1]
    y = _conv(h, conv, batch, length)
    return _GroupNormMish.apply(y, gn.weight, gn.bias, batch, length, gn.num_groups, gn.eps)


def _resblock(rb, h, memb, batch: int, length: int, film=None):
    """ResidualBlock (reference jannerunet.py:51-69): CNA2(CNA1(x) + Linear(Mish(emb))) + skip(x).  `memb` = Mish(emb), evaluated once
    for all blocks (every block's emb_mlp starts with the same Mish); the block's Linear is a library node -- `film`: its output when
    the caller ran all blocks' Linears as one node (`_films`)."""
    c_out = rb.conv1[0].out_channels
    lin = rb.emb_mlp[1]
    if film is None:
        film = _LinearMish.apply(memb, lin.weight, lin.bias, False)                  # (batch, c_out)
    (conv1, gn1), (conv2, gn2), rc = rb.conv1[:2], rb.conv2[:2], rb.residual_conv
    if os.environ.get("CDX_TRAIN_FUSED_ADDS", "1") == "0":                          # (the round-5 graph: one ATen add per term)
        res = h if isinstance(rc, nn.Identity) else _conv(h, rc, batch, length)
        a1 = _cna(h, rb.conv1, batch, length)
        a1 = (a1.view(batch, length, c_out) + film[:, None, :]).view(batch * length, c_out)
        return _cna(a1, rb.conv2, batch, length) + res
    pair = conv1.stride[0] == 1 and (isinstance(rc, nn.Identity) or (isinstance(rc, nn.Conv1d) and rc.kernel_size[0] == 1 and rc.stride[0] == 1
                                                                     and rc.padding[0] == 0))
    if pair and os.environ.get("CDX_TRAIN_CONV_PAIR", "1") != "0":
        ident = isinstance(rc, nn.Identity)
        y1, res = _ConvPair.apply(h, conv1.weight, conv1.bias, None if ident else rc.weight, None if ident else rc.bias, batch, length,
                                  conv1.padding[0])
    else:
        res = h if isinstance(rc, nn.Identity) else _conv(h, rc, batch, length)
        y1 = _conv(h, conv1, batch, length)
    a1 = _GroupNormMishAdd.apply(y1, gn1.weight, gn1.bias, film, None, batch, length, gn1.num_groups, gn1.eps)
    return _GroupNormMishAdd.apply(_conv(a1, conv2, batch, length), gn2.weight, gn2.bias, None, res, batch, length, gn2.num_groups, gn2.eps)


@_with_weight_packs
def janner_forward(net, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor]) -> torch.Tensor:
    """``JannerUNet1d.forward`` (reference nn_diffusion/jannerunet.py:154-201) with autograd, every convolution and GroupNorm on the
    library's kernels.  x (b, H, D) -> (b, H, D)."""
    b, length, d = x.shape
    emb = net.map_noise(noise)
    if condition is not None:                              # (the reference adds zeros otherwise, jannerunet.py:183: the same numbers)
        emb = emb + condition
    emb = torch.nn.functional.mish(_sequential(net.map_emb, emb.contiguous()))        # Mish(emb): what every block's emb_mlp starts with
    h = x.reshape(b * length, d)
    skips = []
    films = _films([rb for lvl in net.downs for rb in lvl[:2]] + [net.mid_block1, net.mid_block2] + [rb for lvl in net.ups for rb in lvl[:2]], emb)
    block = lambda rb, h, length: _resblock(rb, h, emb, b, length, films.get(id(rb)))
    for res1, res2, _, down in net.downs:
        h = block(res2, block(res1, h, length), length)
        skips.append((h, length))
        if not isinstance(down, nn.Identity):
            h = _conv(h, down.conv, b, length)
            length = (length - 1) // 2 + 1
    h = block(net.mid_block2, block(net.mid_block1, h, length), length)
    for res1, res2, _, up in net.ups:
        skip, l_skip = skips.pop()
        assert l_skip == length
        h = torch.cat([h, skip], dim=1)
        h = block(res2, block(res1, h, length), length)
        if not isinstance(up, nn.Identity):
            h = _ConvT.apply(h, up.conv.weight, up.conv.bias, b, length)
            length *= 2
    fc = net.final_conv
    h = _cna(h, fc, b, length)
    h = _conv(h, fc[3], b, length)
    return h.view(b, length, d)


def supports_half_janner(net, x: torch.Tensor, condition=None) -> bool:
    """HalfJannerUNet1d (GroupNorm) with fp32 parameters on a ROCm device, called with autograd on: the classifier's training forward
    (``CumRewClassifier.update`` / ``update_classifier``, reference classifier/base.py:47-58, diffusionsde.py:143-149)."""
    from ..nn_classifier.half_jannerunet import HalfJannerUNet1d
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and type(net) is HalfJannerUNet1d):
        return False
    if net.norm_type != "groupnorm" or net.kernel_size % 2 == 0 or net.kernel_size > 5 or x.dim() != 3 or x.shape[1] != net.horizon:
        return False
    if not _groupnorms_ok(net) or not _wants_grad(net, x, condition):
        return False
    length = net.horizon                       # the backward of a stride-2 conv is written for even input lengths (short horizons: ATen)
    for down in [d for _, _, d in net.downs] + [net.mid_block1[1], net.mid_block2[1]]:
        if not isinstance(down, nn.Identity):
            if length % 2:
                return False
            length //= 2
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


@_with_weight_packs
def half_janner_forward(net, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor]) -> torch.Tensor:
    """``HalfJannerUNet1d.forward`` (reference nn_classifier/half_jannerunet.py:52-63) with autograd, every convolution, GroupNorm and
    Linear on the library's kernels.  x (b, H, D) -> (b, out_dim)."""
    b, length, d = x.shape
    emb = net.map_noise(noise)
    if condition is not None:
        emb = emb + condition
    raw = _sequential(net.map_emb, emb.contiguous())       # the head concatenates map_emb's output as it is ...
    memb = torch.nn.functional.mish(raw)                   # ... every block's emb_mlp starts with its Mish
    h = x.reshape(b * length, d)
    films = _films([rb for lvl in net.downs for rb in lvl[:2]] + [net.mid_block1[0], net.mid_block2[0]], memb)
    for res1, res2, down in net.downs:
        h = _resblock(res2, _resblock(res1, h, memb, b, length, films.get(id(res1))), memb, b, length, films.get(id(res2)))
        if not isinstance(down, nn.Identity):
            h = _conv(h, down.conv, b, length)
            length = (length - 1) // 2 + 1
    for block, down in (net.mid_block1, net.mid_block2):
        h = _conv(_resblock(block, h, memb, b, length, films.get(id(block))), down.conv, b, length)
        length = (length - 1) // 2 + 1
    # x.flatten(1) of the reference's (b, C, L) layout: channel-major
    flat = h.view(b, length, -1).permute(0, 2, 1).reshape(b, -1)
    return _sequential(net.final_block, torch.cat([flat, raw], dim=-1))


def supports_chi(net, x: torch.Tensor, condition=None) -> bool:
    """ChiUNet1d with a global condition (the dp_* configuration, BASELINE config 3) with fp32 parameters on a ROCm device, called
    with autograd on."""
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and type(net).__name__ == "ChiUNet1d"):
        return False
    if not net.obs_as_global_cond or condition is None or net.final_conv[0].kernel_size[0] > 5:
        return False
    if not _groupnorms_ok(net) or not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


def _chi_block(rb, h, memb, batch: int, length: int):
    """ChiResidualBlock (reference nn_diffusion/chiunet.py:13-45): CNA2(FiLM(CNA1(x), Linear(Mish(emb)))) + skip(x).  `memb` = Mish(emb),
    evaluated once for all blocks (every block's cond_encoder starts with the same Mish); the block's Linear is a library node."""
    c_out = rb.out_dim
    a1 = _cna(h, rb.conv1, batch, length).view(batch, length, c_out)
    lin = rb.cond_encoder[1]
    film = _LinearMish.apply(memb, lin.weight, lin.bias, False)
    if rb.cond_predict_scale:
        film = film.view(batch, 2, c_out)
        a1 = film[:, 0, None, :] * a1 + film[:, 1, None, :]
    else:
        a1 = a1 + film[:, None, :]
    a2 = _cna(a1.reshape(batch * length, c_out), rb.conv2, batch, length)
    res = h if isinstance(rb.residual_conv, nn.Identity) else _conv(h, rb.residual_conv, batch, length)
    return a2 + res


@_with_weight_packs
def chi_forward(net, x: torch.Tensor, noise: torch.Tensor, condition: torch.Tensor) -> torch.Tensor:
    """``ChiUNet1d.forward`` with a global condition (reference nn_diffusion/chiunet.py:152-192) with autograd, every convolution,
    GroupNorm and FiLM Linear on the library's kernels.  x (b, Ta, act_dim), condition (b, To, obs_dim) -> (b, Ta, act_dim)."""
    b, length, d = x.shape
    emb = net.map_emb(net.map_noise(noise))
    emb = torch.cat([emb, net.global_cond_encoder(torch.flatten(condition, 1))], dim=-1)
    memb = torch.nn.functional.mish(emb)
    h = x.reshape(b * length, d)
    skips = []
    for res1, res2, down in net.downs:
        h = _chi_block(res2, _chi_block(res1, h, memb, b, length), memb, b, length)
        skips.append((h, length))
        if not isinstance(down, nn.Identity):
            h = _conv(h, down.conv, b, length)
            length = (length - 1) // 2 + 1
    for mid in net.mids:
        h = _chi_block(mid, h, memb, b, length)
    for res1, res2, up in net.ups:
        skip, l_skip = skips.pop()
        assert l_skip == length
        h = torch.cat([h, skip], dim=1)
        h = _chi_block(res2, _chi_block(res1, h, memb, b, length), memb, b, length)
        if not isinstance(up, nn.Identity):
            h = _ConvT.apply(h, up.conv.weight, up.conv.bias, b, length)
            length *= 2
    fc = net.final_conv
    h = _cna(h, fc, b, length)
    h = _conv(h, fc[3], b, length)
    return h.view(b, length, d)


# --------------------------------------------------------------------------------------------------------------------- #
# Linear (+ Mish) nodes: the MLP denoisers under autograd -- DQL's policy update back-propagates through sample()           #
# --------------------------------------------------------------------------------------------------------------------- #
class _LinearMish(torch.autograd.Function):
    """y = [Mish](x W^T + b) on (batch, features) rows: forward = ``cdx_gemm_f32`` (bias in the epilogue) [+ ``cdx_act_f32``]; backward:
    dz = dy * Mish'(z) (``cdx_act_bwd_f32``), dx = dz W (the same GEMM on the transposed weight), dW / db = ``cdx_conv_wgrad_f32`` with
    one tap (a TN GEMM over the batch rows) and its bias column sums."""

    @staticmethod
    def forward(ctx, x, weight, bias, mish):
        x = x.contiguous()
        z = _linear(x, weight, bias)
        ctx.mish = mish
        ctx.params, ctx.packs = (weight, bias), _current_packs()
        ctx.save_for_backward(x, weight, z if mish else x.new_empty(0))
        return blocks.activation(z, "mish") if mish else z

    @staticmethod
    def backward(ctx, dy):
        x, weight, z = ctx.saved_tensors
        dz = dy.contiguous()
        if ctx.mish:
            dz = blocks.activation_backward(z, dz, "mish")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _linear(dz, _pack("linear_t", weight, ctx.packs))
        dw, db = _weight_grads(dz, x, x.shape[0], 1, 1, 1, 1, 0, ctx.params[0], ctx.params[1], ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, None


class _LinearAct(torch.autograd.Function):
    """y = act(x W^T + b), act in {None, "mish", "gelu_tanh", ...}: ``cdx_gemm_f32`` with the bias in its epilogue (+ ``cdx_act_f32``
    when the pre-activation must be kept); backward as ``_LinearMish``."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        z = _linear(x, weight, bias)
        ctx.act = act
        ctx.params, ctx.packs = (weight, bias), _current_packs()
        ctx.save_for_backward(x, weight, z if act else x.new_empty(0))
        return blocks.activation(z, act) if act else z

    @staticmethod
    def backward(ctx, dy):
        x, weight, z = ctx.saved_tensors
        dz = dy.contiguous()
        if ctx.act:
            dz = blocks.activation_backward(z, dz, ctx.act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _linear(dz, _pack("linear_t", weight, ctx.packs))
        dw, db = _weight_grads(dz, x, x.shape[0], 1, 1, 1, 1, 0, ctx.params[0], ctx.params[1], ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, None


class _LayerNormAffine(torch.autograd.Function):
    """nn.LayerNorm with gain and shift on (rows, C): ``cdx_layernorm_f32`` / ``cdx_layernorm_bwd_f32`` (+ column sums)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.params = (gamma, beta)
        return blocks.layernorm(x, gamma=gamma, beta=beta, eps=eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dyx = blocks.layernorm_backward(dy, x, gamma=gamma.detach(), eps=ctx.eps, want_dyxhat=True)
        dg = db = None
        if ctx.needs_input_grad[1]:
            sg = _grad_slot(ctx.params[0])
            dg = blocks.colsum(dyx, out=sg)
            _written(sg)
            dg = None if sg is not None else dg
        if ctx.needs_input_grad[2]:
            sb = _grad_slot(ctx.params[1])
            db = blocks.colsum(dy, out=sb)
            _written(sb)
            db = None if sb is not None else db
        return dx, dg, db, None


class _LayerNormMod(torch.autograd.Function):
    """adaLN: LayerNorm(x) (no affine) * (1 + scale[b]) + shift[b] over `tokens` rows per sample (reference dit.py:10-11,33-35,48).
    scale / shift: (B, C) views of one modulation tensor (same row stride)."""

    @staticmethod
    def forward(ctx, x, scale, shift, tokens, eps):
        x = x.contiguous()
        assert scale.stride(0) == shift.stride(0) and scale.stride(1) == shift.stride(1) == 1
        ctx.save_for_backward(x, scale)
        ctx.geom = (tokens, eps)
        return blocks.layernorm(x, scale=scale, shift=shift, rows_per_mod=tokens, eps=eps)

    @staticmethod
    def backward(ctx, dy):
        x, scale = ctx.saved_tensors
        tokens, eps = ctx.geom
        dy = dy.contiguous()
        dx, dyx = blocks.layernorm_backward(dy, x, scale=scale.detach(), rows_per_mod=tokens, eps=eps, want_dyxhat=True)
        b, c = x.shape[0] // tokens, x.shape[1]
        return dx, dyx.view(b, tokens, c).sum(1), dy.view(b, tokens, c).sum(1), None, None


class _Attention(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v per (sample, head) on packed qkv rows: ``cdx_attention_f32`` / ``cdx_attention_bwd_f32``."""

    @staticmethod
    def forward(ctx, qkv, batch, tokens, n_heads):
        qkv = qkv.contiguous()
        ctx.save_for_backward(qkv)
        ctx.geom = (batch, tokens, n_heads)
        return blocks.attention(qkv, batch, tokens, n_heads)

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        return blocks.attention_backward(qkv, dout.contiguous(), *ctx.geom), None, None, None


class _MHA(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh) + mask) -> dropout -> . v per (sample, head), training forward / backward
    (``cdx_mha_train_fwd_f32`` / ``_bwd_f32``).  Operands: ONE packed tensor [q | k | v] (rows, 3 dm) (`kv` None: self-attention), or
    q (batch * Tq, dm) next to a packed [k | v] (batch * Tk, 2 dm) (cross-attention on a memory).  `mask`: additive (Tq, Tk) or None;
    `keep`: the attention-dropout mask (batch, heads, Tq, Tk) of 0 | 1 / (1 - p) or None (``draw_keep``).  Gradients come back packed
    the way the operands were."""

    @staticmethod
    def forward(ctx, q, kv, batch, n_heads, mask, keep):
        q = q.contiguous()
        kv = None if kv is None else kv.contiguous()
        ctx.save_for_backward(q, kv if kv is not None else q.new_empty(0), mask if mask is not None else q.new_empty(0),
                              keep if keep is not None else q.new_empty(0))
        ctx.geom = (batch, n_heads, kv is None, mask is not None, keep is not None)
        qq, kk, vv = _MHA._blocks(q, kv)
        return blocks.mha_train(qq, kk, vv, batch, n_heads, mask=mask, keep=keep)

    @staticmethod
    def _blocks(q, kv):
        if kv is None:
            dm = q.shape[1] // 3
            return q[:, :dm], q[:, dm:2 * dm], q[:, 2 * dm:]
        dm = q.shape[1]
        return q, kv[:, :dm], kv[:, dm:]

    @staticmethod
    def backward(ctx, dout):
        q, kv, mask, keep = ctx.saved_tensors
        batch, n_heads, packed, has_mask, has_keep = ctx.geom
        kv = None if packed else kv
        dq = torch.empty_like(q)
        dkv = None if packed else torch.empty_like(kv)
        blocks.mha_train(*_MHA._blocks(q, kv), batch, n_heads, mask=mask if has_mask else None, keep=keep if has_keep else None,
                         dout=dout.contiguous(), grads=_MHA._blocks(dq, dkv))
        return dq, dkv, None, None, None, None


def draw_keep(p: float, training: bool, batch: int, n_heads: int, tq: int, tk: int, device) -> Optional[torch.Tensor]:
    """The attention-dropout mask of one nn.MultiheadAttention call as the kernel wants it -- 0 | 1 / (1 - p) per (sample, head, query,
    key), one Bernoulli draw from torch's device generator (so ``torch.manual_seed`` and HIP-graph replay treat it like the noise /
    timestep draws of the step) -- or None when the module does not drop (eval mode or p == 0)."""
    if not training or p <= 0.0:
        return None
    return torch.empty((batch, n_heads, tq, tk), device=device, dtype=torch.float32).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))


def supports_idql(net, x: torch.Tensor, condition=None) -> bool:
    """IDQLMlp / NewIDQLMlp (BASELINE config 5's SynthER residual MLP) with autograd on, on a ROCm device."""
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if type(net).__name__ not in ("IDQLMlp", "NewIDQLMlp") or net.affine_in.out_features > 4096 or not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


@_with_weight_packs
def idql_forward(net, x, noise, condition):
    """``IDQLMlp.forward`` (reference nn_diffusion/idqlmlp.py:9-49): x + Linear(Mish(Linear(LayerNorm(Dropout(x))))) blocks; every Linear
    and LayerNorm a library node, nn.Dropout (an RNG draw) and the feature concat stay ATen."""
    h = _LinearAct.apply(net._features(x, noise, condition), net.affine_in.weight, net.affine_in.bias, None)
    for rb in net.ln_resnet:
        drop, ln, l1, _, l2 = rb.net
        z = _LayerNormAffine.apply(drop(h), ln.weight, ln.bias, ln.eps)
        z = _LinearAct.apply(z, l1.weight, l1.bias, "mish")
        h = h + _LinearAct.apply(z, l2.weight, l2.bias, None)
    head = net.affine_out
    if isinstance(head, nn.Sequential):                    # NewIDQLMlp: Mish before the output projection
        h, head = head[0](h), head[1]
    return _LinearAct.apply(h, head.weight, head.bias, None)


def supports_dit(net, x: torch.Tensor, condition=None) -> bool:
    """DiT1d (BASELINE config 4) with autograd on, on a ROCm device: <= 64 tokens, head_dim <= 64."""
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and
            type(net).__name__ in ("DiT1d", "HalfDiT1d")):       # (HalfDiT1d: the same trunk with a d_model / 2 wide final layer)
        return False
    blk = net.blocks[0] if len(net.blocks) else None
    if blk is None or x.shape[1] > 64 or net.d_model // blk.attn.num_heads > 64 or net.d_model > 4096 or net.d_model % blk.attn.num_heads:
        return False
    if not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


@_with_weight_packs
def dit_forward(net, x, noise, condition):
    """``DiT1d.forward`` (reference nn_diffusion/dit.py:10-50,108-130) with autograd: x_proj / qkv / out_proj / fc1(+GELU) / fc2 / head
    GEMMs, LayerNorm + adaLN modulate and the attention core (with its dropout mask in train mode) on library nodes, forward and
    backward; the (batch, d) embedding / modulation Linears, the gates, the residual adds and the MLP's nn.Dropout stay ATen.  The block keeps the reference's quirk: the residual
    stream continues from the MODULATED LayerNorm output."""
    b, tokens, d_in = x.shape
    d = net.d_model
    if net.pos_emb_cache is None or net.pos_emb_cache.shape[0] != tokens:
        net.pos_emb_cache = net.pos_emb(torch.arange(tokens, device=x.device))
    emb = net._embed(noise, condition)
    h = _LinearAct.apply(x.reshape(b * tokens, d_in), net.x_proj.weight, net.x_proj.bias, None)
    h = (h.view(b, tokens, d) + net.pos_emb_cache[None]).view(b * tokens, d)
    for blk in net.blocks:
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = blk.adaLN_modulation(emb).chunk(6, dim=1)
        att = blk.attn
        h = _LayerNormMod.apply(h, sc_a, sh_a, tokens, blk.norm1.eps)
        qkv = _LinearAct.apply(h, att.in_proj_weight, att.in_proj_bias, None)
        keep = draw_keep(att.dropout, att.training, b, att.num_heads, tokens, tokens, x.device)
        core = _Attention.apply(qkv, b, tokens, att.num_heads) if keep is None else _MHA.apply(qkv, None, b, att.num_heads, None, keep)
        o = _LinearAct.apply(core, att.out_proj.weight, att.out_proj.bias, None)
        h = (h.view(b, tokens, d) + g_a[:, None, :] * o.view(b, tokens, d)).view(b * tokens, d)
        m = _LayerNormMod.apply(h, sc_m, sh_m, tokens, blk.norm2.eps)
        f = blk.mlp[2](_LinearAct.apply(m, blk.mlp[0].weight, blk.mlp[0].bias, "gelu_tanh"))       # (nn.Dropout: an ATen draw in train mode)
        f = _LinearAct.apply(f, blk.mlp[3].weight, blk.mlp[3].bias, None)
        h = (h.view(b, tokens, d) + g_m[:, None, :] * f.view(b, tokens, d)).view(b * tokens, d)
    fl = net.final_layer
    shift, scale = fl.adaLN_modulation(emb).chunk(2, dim=1)
    m = _LayerNormMod.apply(h, scale, shift, tokens, fl.norm_final.eps)
    return _LinearAct.apply(m, fl.linear.weight, fl.linear.bias, None).view(b, tokens, fl.linear.out_features)


def _mha_ok(att: nn.MultiheadAttention, d: int) -> bool:
    return (att._qkv_same_embed_dim and att.batch_first and att.in_proj_bias is not None and att.bias_k is None and not att.add_zero_attn and
            att.embed_dim == d and d % att.num_heads == 0 and d // att.num_heads <= 64)


def _tf_layer_ok(layer, d: int, decoder: bool) -> bool:
    import torch.nn.functional as F
    if not (layer.norm_first and layer.activation is F.gelu and _mha_ok(layer.self_attn, d)):
        return False
    return not decoder or _mha_ok(layer.multihead_attn, d)


def supports_chitf(net, x: torch.Tensor, condition=None) -> bool:
    """ChiTransformer (Diffusion Policy's transformer denoiser, the dp_* pipelines) with autograd on, on a ROCm device: the shapes its
    masks were built for (Ta action tokens, 1 + To memory tokens, both <= 64), head_dim <= 64, pre-norm GELU layers as the class
    constructs them (reference nn_diffusion/chitransformer.py:90-135)."""
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and
            type(net).__name__ == "ChiTransformer"):
        return False
    d = net.act_emb.out_features
    if x.shape[1] != net.T or net.T > 64 or net.T_cond > 64 or d > 4096 or net.obs_emb is None:
        return False
    if condition is not None and not (torch.is_tensor(condition) and condition.dim() == 3 and condition.shape[1] == net.To and
                                      condition.dtype == torch.float32):
        return False
    enc = net.encoder
    if isinstance(enc, nn.TransformerEncoder):
        if enc.norm is not None or not all(_tf_layer_ok(l, d, False) for l in enc.layers):
            return False
    elif not (isinstance(enc, nn.Sequential) and len(enc) == 3 and isinstance(enc[1], nn.Mish)):
        return False
    if net.decoder.norm is not None or not all(_tf_layer_ok(l, d, True) for l in net.decoder.layers) or not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


def _self_attention(att, h2, batch, tokens, mask):
    qkv = _LinearAct.apply(h2, att.in_proj_weight, att.in_proj_bias, None)
    keep = draw_keep(att.dropout, att.training, batch, att.num_heads, tokens, tokens, h2.device)
    core = _Attention.apply(qkv, batch, tokens, att.num_heads) if mask is None and keep is None else \
        _MHA.apply(qkv, None, batch, att.num_heads, mask, keep)
    return _LinearAct.apply(core, att.out_proj.weight, att.out_proj.bias, None)


def _ffn(layer, h2):
    f = layer.dropout(_LinearAct.apply(h2, layer.linear1.weight, layer.linear1.bias, "gelu"))
    return _LinearAct.apply(f, layer.linear2.weight, layer.linear2.bias, None)


@_with_weight_packs
def chitf_forward(net, x, noise, condition):
    """``ChiTransformer.forward`` (reference nn_diffusion/chitransformer.py:137-158) with autograd, as nn.TransformerEncoder /
    nn.TransformerDecoder run it in TRAIN mode (pre-norm layers: x += drop(attn(norm(x))), x += drop(ffn(norm(x)))): every Linear, every
    LayerNorm and the three attention cores (memory encoder self-attention, causal self-attention, staggered memory cross-attention --
    each with its dropout mask) on library nodes, forward and backward; token concat, positional adds, residual adds and the
    nn.Dropout modules (RNG draws) stay ATen."""
    b, ta, act_dim = x.shape
    d, tc = net.act_emb.out_features, net.T_cond
    if condition is None:
        condition = torch.zeros((b, net.To, net.obs_dim), device=x.device)
    obs = _LinearAct.apply(condition.reshape(b * net.To, net.obs_dim), net.obs_emb.weight, net.obs_emb.bias, None)
    mem = torch.cat([net.map_noise(noise).unsqueeze(1), obs.view(b, net.To, d)], dim=1)
    mem = net.drop(mem + net.cond_pos_emb[:, :tc, :]).reshape(b * tc, d)
    enc = net.encoder
    if isinstance(enc, nn.Sequential):
        mem = _LinearAct.apply(_LinearAct.apply(mem, enc[0].weight, enc[0].bias, "mish"), enc[2].weight, enc[2].bias, None)
    else:
        for layer in enc.layers:
            n1, n2 = layer.norm1, layer.norm2
            mem = mem + layer.dropout1(_self_attention(layer.self_attn, _LayerNormAffine.apply(mem, n1.weight, n1.bias, n1.eps), b, tc, None))
            mem = mem + layer.dropout2(_ffn(layer, _LayerNormAffine.apply(mem, n2.weight, n2.bias, n2.eps)))
    h = _LinearAct.apply(x.reshape(b * ta, act_dim), net.act_emb.weight, net.act_emb.bias, None)
    h = net.drop(h.view(b, ta, d) + net.pos_emb[:, :ta, :]).reshape(b * ta, d)
    causal, staggered = net.mask.detach(), net.memory_mask.detach()
    for layer in net.decoder.layers:
        n1, n2, n3, ca = layer.norm1, layer.norm2, layer.norm3, layer.multihead_attn
        h = h + layer.dropout1(_self_attention(layer.self_attn, _LayerNormAffine.apply(h, n1.weight, n1.bias, n1.eps), b, ta, causal))
        q = _LinearAct.apply(_LayerNormAffine.apply(h, n2.weight, n2.bias, n2.eps), ca.in_proj_weight[:d], ca.in_proj_bias[:d], None)
        kv = _LinearAct.apply(mem, ca.in_proj_weight[d:], ca.in_proj_bias[d:], None)
        keep = draw_keep(ca.dropout, ca.training, b, ca.num_heads, ta, tc, x.device)
        o = _LinearAct.apply(_MHA.apply(q, kv, b, ca.num_heads, staggered, keep), ca.out_proj.weight, ca.out_proj.bias, None)
        h = h + layer.dropout2(o)
        h = h + layer.dropout3(_ffn(layer, _LayerNormAffine.apply(h, n3.weight, n3.bias, n3.eps)))
    h = _LayerNormAffine.apply(h, net.ln_f.weight, net.ln_f.bias, net.ln_f.eps)
    return _LinearAct.apply(h, net.head.weight, net.head.bias, None).view(b, ta, net.head.out_features)


def supports_mlp(net, x: torch.Tensor, condition=None) -> bool:
    """DQLMlp / DVInvMlp (Linear -> Mish trunks) with fp32 parameters on a ROCm device, called with autograd on -- what
    ``sample(..., requires_grad=True)`` of the Diffusion-QL policy update runs at every denoising step (reference
    pipelines/dql_d4rl_mujoco.py:101, diffusionsde.py:401-427)."""
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if type(net).__name__ not in ("DQLMlp", "DVInvMlp") or not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


def _sequential(seq: nn.Sequential, h):
    """Linear [-> Mish] chains of an nn.Sequential on the library's kernels (anything else: the module itself)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear) and m.bias is not None:
            mish = i + 1 < len(mods) and isinstance(mods[i + 1], nn.Mish)
            h = _LinearMish.apply(h, m.weight, m.bias, mish)
            i += 2 if mish else 1
        else:
            h = m(h)
            i += 1
    return h


# --------------------------------------------------------------------------------------------------------------------- #
# Row-wise MLP chains under autograd: the heads trained NEXT to a denoiser                                                     #
# --------------------------------------------------------------------------------------------------------------------- #
# The critics of Diffusion-QL / IDQL (utils/critics.py: Linear -> LayerNorm -> Tanh / Mish towers; reference utils/iql.py:40-95,
# building_blocks.py:111-147) and the inverse-dynamics heads (invdynamic/mlp.py:7-293) are trained by their own ``update`` methods or by the
# pipelines' loops.  With autograd on, on a ROCm device, their nn.Sequential chains run on the same library nodes as the denoisers'
# Linears (cdx_gemm_f32 / cdx_act_f32 / cdx_layernorm_f32 forward, cdx_gemm_f32 / cdx_conv_wgrad_f32 / cdx_layernorm_bwd_f32 backward).
_CHAIN_ACTS = {nn.Mish: "mish", nn.ReLU: "relu", nn.Tanh: "tanh", nn.SiLU: "silu"}


def _chain_act(m) -> Optional[str]:
    if type(m) in _CHAIN_ACTS:
        return _CHAIN_ACTS[type(m)]
    if type(m) is nn.GELU:
        return "gelu_tanh" if m.approximate == "tanh" else "gelu"
    return None


def _chain_plan(seq):
    """[(kind, module, activation)] of a Sequential this path understands, or None: Linear (+ activation), LayerNorm (+ activation),
    a lone activation, Identity, inactive Dropout."""
    def flat(q):                               # (utils.Mlp nests one Sequential(Linear, activation) per hidden layer)
        for m in q:
            if isinstance(m, nn.Sequential):
                yield from flat(m)
            else:
                yield m
    mods = [m for m in flat(seq) if not isinstance(m, nn.Identity) and not (isinstance(m, nn.Dropout) and (not m.training or m.p == 0.0))]
    plan, i = [], 0
    while i < len(mods):
        m = mods[i]
        act = _chain_act(mods[i + 1]) if i + 1 < len(mods) else None
        if type(m) is nn.Linear and m.bias is not None:
            plan.append(("linear", m, act))
        elif type(m) is nn.LayerNorm and m.elementwise_affine and len(m.normalized_shape) == 1 and m.bias is not None:
            plan.append(("norm", m, act))
        elif _chain_act(m) is not None:
            plan.append(("act", m, _chain_act(m)))
            act = None
        else:
            return None
        i += 2 if act is not None and plan[-1][0] != "act" else 1
    return plan


def supports_chain(seq, x: torch.Tensor) -> bool:
    if not (enabled() and torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if not isinstance(seq, nn.Sequential) or not (x.requires_grad or any(p.requires_grad for p in seq.parameters())):
        return False
    if not all(p.dtype == torch.float32 and p.is_cuda for p in seq.parameters()):
        return False
    return _chain_plan(seq) is not None


@_with_weight_packs
def chain_forward(seq, x: torch.Tensor) -> torch.Tensor:
    """``seq(x)`` for (rows, features) with autograd, every Linear / LayerNorm / activation a library node (see above)."""
    h = x
    for kind, m, act in _chain_plan(seq):
        if kind == "linear":
            h = _LinearAct.apply(h, m.weight, m.bias, act)
        elif kind == "norm":
            h = _LayerNormAffine.apply(h, m.weight, m.bias, m.eps)
            if act is not None:
                h = _Act.apply(h, act)
        else:
            h = _Act.apply(h, act)
    return h


@_with_weight_packs
def dql_forward(net, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor]) -> torch.Tensor:
    """``DQLMlp.forward`` / ``DVInvMlp.forward`` (reference nn_diffusion/dqlmlp.py:30-52, dvinvmlp.py:30-47) with autograd, every Linear
    and Mish on the library's kernels: features = [x | time_mlp(map_noise(t)) | condition], trunk, head."""
    if condition is None:
        condition = torch.zeros(x.shape[0], net.obs_dim, device=x.device)
    temb = _sequential(net.time_mlp, net.map_noise(noise).contiguous())
    h = torch.cat([x, temb, condition], -1)
    h = _sequential(net.mid_layer, h)
    return _LinearMish.apply(h, net.final_layer.weight, net.final_layer.bias, False)


def supports_pearce(net, x: torch.Tensor, condition=None) -> bool:
    """PearceMlp (BASELINE config 1, the dbc_* pipelines) with autograd on, on a ROCm device."""
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and
            type(net).__name__ == "PearceMlp"):
        return False
    if not _groupnorms_ok(net) or not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


def _fc_block(block, h):
    """FCBlock (reference nn_diffusion/pearcemlp.py:10-23): Linear -> GroupNorm1d over the features of one row -> GELU(erf)."""
    lin, gn, _ = block.model
    z = _LinearAct.apply(h, lin.weight, lin.bias, None)
    z = _GroupNormMish.apply(z, gn.weight, gn.bias, z.shape[0], 1, gn.num_groups, gn.eps, "none")
    return _Act.apply(z, "gelu")


@_with_weight_packs
def pearce_forward(net, x, noise, condition):
    """``PearceMlp.forward`` (reference nn_diffusion/pearcemlp.py:26-77) with autograd: every Linear, GroupNorm1d and GELU / LeakyReLU
    on library nodes; the feature concats, the skip scaling and the residual adds stay ATen."""
    if condition is None:
        condition = torch.zeros(x.shape[0], net.To, net.emb_dim, device=x.device)
    t = noise.unsqueeze(-1)
    e0, _, e2 = net.act_emb
    a = _LinearAct.apply(_LinearAct.apply(x, e0.weight, e0.bias, "leaky"), e2.weight, e2.bias, None)
    h = _fc_block(net.fcs[0], torch.cat([a, net.map_noise(noise), torch.flatten(condition, 1)], -1))
    for block in (net.fcs[1], net.fcs[2]):
        skip = h / net.SKIP_SCALE
        h = _fc_block(block, torch.cat([skip, x, t], -1)) + skip
    last = net.fcs[3]
    return _LinearAct.apply(torch.cat([h, x, t], -1), last.weight, last.bias, None)


def supports_sfbc(net, x: torch.Tensor, condition=None) -> bool:
    """SfBCUNet (the sfbc_* pipelines' residual MLP) with autograd on, on a ROCm device."""
    if not (enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and
            type(net).__name__ == "SfBCUNet") or not _wants_grad(net, x, condition):
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in net.parameters())


def _sfbc_block(rb, h, c):
    """SiLU(L2(SiLU(L1 x) + Lc c)) + skip(x) (reference nn_diffusion/sfbc_unet.py:9-24)."""
    l1, l2, lc = rb.linear1[0], rb.linear2[0], rb.linearc
    z = _LinearAct.apply(h, l1.weight, l1.bias, "silu") + _LinearAct.apply(c, lc.weight, lc.bias, None)
    z = _LinearAct.apply(z, l2.weight, l2.bias, "silu")
    return z + (h if isinstance(rb.skip, nn.Identity) else _LinearAct.apply(h, rb.skip.weight, rb.skip.bias, None))


@_with_weight_packs
def sfbc_forward(net, x, noise, condition):
    """``SfBCUNet.forward`` (reference nn_diffusion/sfbc_unet.py:60-82) with autograd, every Linear (+ SiLU) a library node."""
    t0, _, t2 = net.t_layer
    c = _LinearAct.apply(_LinearAct.apply(net.map_noise(noise).contiguous(), t0.weight, t0.bias, "silu"), t2.weight, t2.bias, None)
    if condition is not None:                              # (the reference adds zeros otherwise: the same numbers)
        c = c + condition
    kept = []
    for block in net.down_blocks:
        x = _sfbc_block(block, x, c)
        kept.append(x)
    x = _sfbc_block(net.mid_block, x, c)
    for block in net.up_blocks:
        x = _sfbc_block(block, torch.cat([x, kept.pop()], dim=-1), c)
    return _LinearAct.apply(x, net.out_layer.weight, net.out_layer.bias, None)


# --------------------------------------------------------------------------------------------------------------------- #
# forward + backward of update() as ONE HIP graph (the default where a probe finds the step capturable; CDX_TRAIN_GRAPH=0 / 1)       #
# --------------------------------------------------------------------------------------------------------------------- #
RECAPTURE_LIMIT = 3
SHAPE_LIMIT = 4            # captured graphs per agent (one per batch shape / train-eval state); further shapes run eagerly


class NotCapturable(RuntimeError):
    pass


class GraphedStep:
    """``loss = agent.loss(x0, condition); loss.backward()`` captured once into a HIP graph and replayed per step.

    update() of config 2 is ~600 autograd nodes: ~14 ms of Python / dispatcher time per step at ANY batch size against ~8 ms of kernel
    time (profiles/r04_update_bench.txt) -- launch-bound, which is what HIP graphs are for.  Everything the LIBRARY does inside the
    captured region is capturable by construction: its launches go to the current (capturing) stream and allocate nothing, torch's
    allocations come from the graph's private pool, the timestep / noise draws of ``add_noise`` use the graph-safe device generator.
    What user code wrapped around ``loss()`` does is not ours to know, so the first warm-up step is a PROBE (`probe=True`): it runs
    eagerly under ``torch.cuda.set_sync_debug_mode("error")`` -- anything that synchronises (a draw from the CPU generator moved to the
    device, ``.item()``, a data-dependent shape) raises there, BEFORE any capture was attempted (a capture that fails midway leaves
    torch's generator / allocator bookkeeping in the capturing state), and the agent keeps the eager path (``NotCapturable``).
    Static buffers: the batch (copied in before each replay), the loss, the parameters' ``.grad`` tensors (allocated by the warm-up
    steps; the captured backward ACCUMULATES into them in place -- the optimiser zeroes them in place after each step).

    Not captured: the optimiser step (its bias-correction scalars change per step and travel as kernel arguments), ``loss.item()``."""

    def __init__(self, agent, x0: torch.Tensor, condition: Optional[torch.Tensor], probe: bool = False):
        """`agent`: anything with ``.model`` (the trained module) and ``.loss(x0, condition)``; `condition` may be a TUPLE of tensors --
        the classifier step's (noise level, target) -- every member a static buffer."""
        dev = x0.device
        self.x0 = x0.detach().clone()
        if isinstance(condition, tuple):
            self.cond = tuple(c.detach().clone() for c in condition)
        else:
            self.cond = None if condition is None else condition.detach().clone()
        params = self.params = [p for p in agent.model.parameters() if p.requires_grad]
        had_grad = [p.grad is not None for p in params]
        # gradients the caller accumulated BEFORE this first update() (an auxiliary loss.backward(), gradient accumulation): the reference's
        # update() adds onto them, so the warm-up / capture below must hand them back exactly
        kept = [p.grad.detach().clone() if p.grad is not None else None for p in params]

        def restore():
            for p, had, g0 in zip(params, had_grad, kept):
                if not had:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.copy_(g0)
                else:
                    p.grad = g0
            torch.cuda.set_rng_state(rng, dev)
            torch.set_rng_state(rng_cpu)
        # the warm-up steps below draw timesteps / noise like any step: put the generators back afterwards, so that the FIRST replay
        # consumes what the first eager step would have (a replay reads the generator's offset at replay time and advances it by what
        # the captured draws consume -- the same numbers an eager step draws from the same state)
        rng, rng_cpu = torch.cuda.get_rng_state(dev), torch.get_rng_state()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(side):
                for i in range(2):                     # warm-up: lazy initialisation, allocator, the .grad tensors
                    if probe and i == 0:
                        mode = torch.cuda.get_sync_debug_mode()
                        torch.cuda.set_sync_debug_mode("error")
                        try:
                            with grads_in_place(params):
                                agent.loss(self.x0, self.cond).backward()
                        finally:
                            torch.cuda.set_sync_debug_mode(mode)
                    else:
                        with grads_in_place(params):
                            agent.loss(self.x0, self.cond).backward()
        except Exception as e:  # noqa: BLE001 -- whatever the probe step tripped over: this agent's step is not ours to capture
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            restore()                                  # nothing of the probe is part of any step
            raise NotCapturable(f"{type(e).__name__}: {e}") from e
        torch.cuda.current_stream(dev).wait_stream(side)
        for p in params:                               # the warm-up gradients are not part of any step
            if p.grad is not None:
                p.grad.zero_()
        # the parameters the step gives a gradient to: a replay marks THOSE as written for the optimiser, no others
        self.written = [p for p in params if p.grad is not None]
        self.graph = torch.cuda.CUDAGraph()
        try:
            # (thread-local error mode: a CUDA call from ANOTHER thread -- a DataLoader's pin_memory thread -- does not fail the capture)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = agent.loss(self.x0, self.cond)
                with grads_in_place(params):                 # (the captured launches add into the static .grad tensors directly)
                    self.loss.backward()
        except Exception as e:  # noqa: BLE001 -- something the probe did not see (it is a prototype check, and blind to pageable H2D copies)
            # torch.cuda.graph's exit has ended the capture; whatever half-built graph exists is dropped, the eager step serves this agent.
            # (What this can repair is a failure ABOVE the driver -- an exception out of user code.  An illegal CUDA call under capture
            #  invalidates the capture inside the driver and torch's capture_end throws before it un-registers its generator / allocator
            #  pool; nothing in Python undoes that, which is why the eager probe runs first.)
            self.graph = None
            torch.cuda.synchronize(dev)
            restore()
            raise NotCapturable(f"capture failed: {type(e).__name__}: {e}") from e
        for p, g0 in zip(params, kept):                # (capture does not run the kernels) back to what the caller had: zero or its own sums
            if p.grad is not None:
                if g0 is not None:
                    p.grad.copy_(g0)
                else:
                    p.grad.zero_()
        torch.cuda.set_rng_state(rng, dev)
        torch.set_rng_state(rng_cpu)
        self.sig = self._signature()

    def _signature(self):
        """The addresses the captured launches read and write: parameters and their gradient tensors."""
        return tuple((p.data_ptr(), -1 if p.grad is None else p.grad.data_ptr()) for p in self.params)

    def valid(self) -> bool:
        """False once a parameter or a ``.grad`` tensor is no longer the one the graph was captured on (``zero_grad(set_to_none=True)``,
        ``module.to(...)``, a re-created parameter): replaying would read or accumulate into memory that is not the model's any more."""
        return self._signature() == self.sig

    def replay(self, x0, condition):
        self.x0.copy_(x0)
        if isinstance(self.cond, tuple):
            for dst, src in zip(self.cond, condition):
                dst.copy_(src)
        elif self.cond is not None:
            self.cond.copy_(condition)
        self.graph.replay()
        return self.loss


def _native_training_net(net, x0, condition) -> bool:
    with torch.enable_grad():
        return (supports(net, x0, condition) or supports_chi(net, x0, condition) or supports_dit(net, x0, condition) or
                supports_chitf(net, x0, condition) or supports_idql(net, x0, condition) or supports_mlp(net, x0, condition) or
                supports_pearce(net, x0, condition) or supports_sfbc(net, x0, condition))


def graphed_step(agent, x0, condition, kwargs) -> Optional[GraphedStep]:
    """The cached GraphedStep of (agent, batch shape), or None (the eager path).  CDX_TRAIN_GRAPH: "auto" (default) -- agents whose
    denoiser the native training path serves (JannerUNet1d, ChiUNet1d, DiT1d, ChiTransformer, IDQLMlp, PearceMlp, SfBCUNet, DQLMlp / DVInvMlp on a ROCm device), no extra
    loss arguments, and whose first step passes the capturability probe (GraphedStep); "1": no probe; "0": never."""
    mode = os.environ.get("CDX_TRAIN_GRAPH", "auto")
    if mode == "0" or kwargs or not torch.is_tensor(x0) or not x0.is_cuda or not torch.is_grad_enabled() or \
            torch.cuda.is_current_stream_capturing():
        return None
    if condition is not None and not (torch.is_tensor(condition) and condition.device == x0.device):
        return None                                        # (dictionaries of observations, host tensors: not a static buffer -- eager)
    if agent.__dict__.get("_cdx_graph_off"):
        return None
    net = agent.model["diffusion"]
    if not _native_training_net(net, x0, condition):       # (the raw condition stands in for the encoded one: only its presence matters)
        return None
    key = (tuple(x0.shape), None if condition is None else tuple(condition.shape), agent.model.training)
    cache = agent.__dict__.setdefault("_cdx_graphed", {})
    g = cache.get(key)
    if g is not None and not g.valid():
        # captured on tensors that are gone: capture again -- unless that keeps happening (an optimiser that drops the gradients after
        # every step): then the eager path serves this agent
        agent._cdx_recaptures = getattr(agent, "_cdx_recaptures", 0) + 1
        g = cache[key] = None
    if getattr(agent, "_cdx_recaptures", 0) > RECAPTURE_LIMIT:
        return None
    if g is None:
        if key not in cache and len(cache) >= SHAPE_LIMIT:
            return None                                # (a loop over ever-changing batch shapes: capturing each would cost more than it saves)
        try:
            g = cache[key] = GraphedStep(agent, x0, condition, probe=(mode != "1"))
        except NotCapturable as e:
            agent.__dict__["_cdx_graph_off"] = str(e)  # (kept for diagnostics: why this agent steps eagerly)
            cache.pop(key, None)
            return None
    return g


class _ClassifierStep:
    """``loss = classifier.loss(x, noise, y)`` in the shape GraphedStep captures: ``.model`` and ``.loss(x0, (noise, y))``."""

    def __init__(self, clf):
        self.clf, self.model = clf, clf.model

    def loss(self, x, cond):
        return self.clf.loss(x, cond[0], cond[1])


def graphed_classifier_step(clf, x, noise, y) -> Optional[GraphedStep]:
    """The cached GraphedStep of a classifier's ``loss(x, noise, y); backward()`` (``BaseClassifier.update``), or None (the eager pair):
    same rules as `graphed_step` -- a HalfJannerUNet1d on a ROCm device, tensors for all three inputs, a first step that passes the
    capturability probe (CDX_TRAIN_GRAPH: "auto" / "1" / "0")."""
    mode = os.environ.get("CDX_TRAIN_GRAPH", "auto")
    if mode == "0" or not all(torch.is_tensor(t) and t.is_cuda and t.device == x.device for t in (x, noise, y)) or \
            not torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing() or clf.__dict__.get("_cdx_graph_off"):
        return None
    if not supports_half_janner(clf.model, x, None):
        return None
    key = (tuple(x.shape), tuple(noise.shape), noise.dtype, tuple(y.shape), clf.model.training)
    cache = clf.__dict__.setdefault("_cdx_graphed", {})
    g = cache.get(key)
    if g is not None and not g.valid():
        clf._cdx_recaptures = getattr(clf, "_cdx_recaptures", 0) + 1
        g = cache[key] = None
    if getattr(clf, "_cdx_recaptures", 0) > RECAPTURE_LIMIT:
        return None
    if g is None:
        if key not in cache and len(cache) >= SHAPE_LIMIT:
            return None
        try:
            g = cache[key] = GraphedStep(_ClassifierStep(clf), x, (noise, y), probe=(mode != "1"))
        except NotCapturable as e:
            clf.__dict__["_cdx_graph_off"] = str(e)
            cache.pop(key, None)
            return None
    return g
