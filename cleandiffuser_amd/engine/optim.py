"""Optimiser side of ``DiffusionModel.update()`` on the gfx950 library (``cdx_optim_f32``, csrc/cdx_optim.hip).

The reference builds ``torch.optim.AdamW(self.model.parameters(), ...)`` (diffusion/basic.py:66) and pipelines keep using that
object (LR schedulers, ``state_dict``), so the drop-in here is a SUBCLASS of ``torch.optim.AdamW``: same constructor, same
``param_groups`` / ``state`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), but on a ROCm device

* ``step(max_norm=..., ema=(model_ema, rate), zero_grad=True)`` is at most three launches for the whole model: the gradient-norm
  pass (two kernels: per-chunk sums, fixed-order reduction -- replaces ``clip_grad_norm_``) and ONE fused pass that clips, applies
  AdamW, folds the new parameter into its EMA copy (reference basic.py:83-86) and leaves the gradient zeroed;
* ``zero_grad()`` keeps the gradient tensors (and so the device pointer table) alive: zeroed in place, by ``step`` for free.
  torch's ``set_to_none=True`` semantics are kept from the optimiser's point of view: a gradient that nobody has written since it was
  zeroed here (its autograd version counter has not moved) counts as ``None`` -- the parameter is not weight-decayed, not
  momentum-stepped and its step count does not advance, exactly as ``torch.optim.AdamW`` skips a parameter without a gradient.

Parameters on the CPU (or options this path does not carry: amsgrad, maximize, non-fp32) take torch's own ``step`` unchanged.
On a ROCm device a missing ``libcdx.so`` is a hard error (runtime.load_library), never a silent fallback.
"""
import ctypes
import math
from typing import Optional

import torch

from . import runtime as R

CHUNK = 4096                      # floats per workgroup (256 threads x 4 float4)
OPT_ADAMW, OPT_EMA, OPT_SUMSQ, OPT_ZERO, OPT_ADAM = 0, 1, 2, 3, 4


class CdxOptimArgs(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("ema", ctypes.c_void_p), ("numel", ctypes.c_void_p), ("chunks", ctypes.c_void_p),
                ("n_tensors", ctypes.c_int32), ("n_chunks", ctypes.c_int32), ("chunk_elems", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("weight_decay", ctypes.c_float), ("step_size", ctypes.c_float), ("bc2_sqrt", ctypes.c_float),
                ("ema_rate", ctypes.c_float), ("max_norm", ctypes.c_float), ("zero_grad", ctypes.c_int32),
                ("partial", ctypes.c_void_p), ("norm", ctypes.c_void_p)]


_declared = False


def _lib():
    global _declared
    lib = R.load_library()
    if not _declared:
        lib.cdx_optim_f32.argtypes = [ctypes.POINTER(CdxOptimArgs), ctypes.c_void_p]
        lib.cdx_optim_f32.restype = ctypes.c_int
        _declared = True
    return lib


class _Table:
    """Device image of the pointer lists of one tensor set: ``ptrs[k]`` = int64 device vector of data pointers of list k, plus the
    shared numel vector and chunk table.  Rebuilt only when a pointer changes (load_state_dict(assign=True), ``.to()``...)."""

    def __init__(self, lists, device):
        self.key = tuple(tuple(t.data_ptr() for t in lst) for lst in lists)
        numel = [t.numel() for t in lists[0]]
        for lst in lists:
            assert [t.numel() for t in lst] == numel
            assert all(t.dtype == torch.float32 and t.is_contiguous() and t.device == lists[0][0].device for t in lst)
        chunks = [(ti, ci) for ti, n in enumerate(numel) for ci in range(-(-n // CHUNK))]
        host = torch.tensor([list(k) for k in self.key] + [numel], dtype=torch.int64)
        self.dev = host.to(device)                                       # rows: one pointer list each, last row = numel
        self.chunks = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(device)
        self.n_tensors, self.n_chunks = len(numel), len(chunks)
        self.partial = torch.empty(max(self.n_chunks, 1), dtype=torch.float32, device=device)
        self.keep = [list(lst) for lst in lists]                         # the tensors must outlive their pointers

    def row(self, k: int) -> int:
        return self.dev.data_ptr() + 8 * k * self.n_tensors

    @property
    def numel_ptr(self) -> int:
        return self.row(self.dev.shape[0] - 1)


def _table(cache: dict, name: str, lists, device) -> _Table:
    key = tuple(tuple(t.data_ptr() for t in lst) for lst in lists)
    hit = cache.get(name)
    if hit is None or hit.key != key:
        hit = cache[name] = _Table(lists, device)
    return hit


def _call(args: CdxOptimArgs, device):
    R._check(_lib().cdx_optim_f32(ctypes.byref(args), R._stream_ptr(device)), "cdx_optim_f32")


def native_device(t: torch.Tensor) -> bool:
    return t.is_cuda and t.dtype == torch.float32


def _bump_versions(tensors):
    """The kernels write parameters behind autograd's back: bump the version counters the native executors' packed-weight caches
    (and autograd's own saved-tensor checks) key on."""
    torch.autograd.graph.increment_version(list(tensors))


def ema_update_native(model: torch.nn.Module, model_ema: torch.nn.Module, rate: float) -> bool:
    """ema <- rate * ema + (1 - rate) * p over every parameter pair in ONE launch (reference basic.py:83-86).  False when the
    parameters are not fp32 tensors on a ROCm device (caller keeps the ATen loop)."""
    ps = [p.detach() for p in model.parameters()]
    es = [p.detach() for p in model_ema.parameters()]
    if not ps or len(ps) != len(es) or not all(native_device(t) and t.is_contiguous() for t in ps + es):
        return False
    dev = ps[0].device
    cache = model_ema.__dict__.setdefault("_cdx_optim_tables", {})
    tab = _table(cache, "ema", [ps, es], dev)
    _call(CdxOptimArgs(p=tab.row(0), ema=tab.row(1), numel=tab.numel_ptr, chunks=tab.chunks.data_ptr(), n_tensors=tab.n_tensors,
                       n_chunks=tab.n_chunks, chunk_elems=CHUNK, mode=OPT_EMA, ema_rate=float(rate)), dev)
    _bump_versions(model_ema.parameters())
    return True


class _FusedStep:
    """The device-side ``step`` / ``zero_grad`` shared by ``FusedAdamW`` and ``FusedAdam`` (a mixin IN FRONT of the torch optimiser
    class: ``super()`` calls reach torch's own methods).  Extra keyword arguments of ``step`` (all optional):

    ``max_norm``  -- clip the global gradient norm first (what ``clip_grad_norm_`` does); the norm lands in ``last_grad_norm``
    ``ema``       -- ``(model, model_ema, rate)``: fold the updated parameters into their EMA copies in the same pass
    ``zero_grad`` -- leave the gradients zeroed (the caller then skips ``zero_grad()``)
    """
    _OPT_MODE = OPT_ADAMW

    def __init__(self, params, **kw):
        super().__init__(params, **kw)
        self._tables = {}
        self._norm = None
        self.last_grad_norm = None
        self._t = {}                # id(parameter) -> step count (python int): the per-parameter `step` TENSORS of torch's state layout
        #                             are refreshed from it only when somebody looks (state_dict), not 196 CPU tensor ops per step
        self._gver = {}             # id(parameter) -> (grad data_ptr, grad._version) right after THIS optimiser zeroed the gradient in
        #                             place: unchanged at the next step = no backward pass wrote it = torch would see `grad is None`

    def _sync_steps(self):
        for group in self.param_groups:
            for p in group["params"]:
                t = self._t.get(id(p))
                if t is not None and p in self.state:
                    self.state[p]["step"].fill_(float(t))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._t = {id(p): int(float(self.state[p]["step"])) for g in self.param_groups for p in g["params"] if p in self.state and "step" in self.state[p]}
        self._tables = {}

    # ---- helpers ----
    @staticmethod
    def _native_group(group) -> bool:
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        ps = [p for p in group["params"] if p.requires_grad]
        return bool(ps) and all(native_device(p) and p.is_contiguous() and
                                (p.grad is None or (not p.grad.is_sparse and p.grad.is_contiguous() and p.grad.dtype == torch.float32))
                                for p in ps)

    def _mode_of(self, group) -> int:
        """Decoupled (AdamW) or L2 (Adam) decay: torch >= 2.6 carries it per group (``decoupled_weight_decay``), else the class says."""
        dec = group.get("decoupled_weight_decay")
        if dec is None:
            return self._OPT_MODE
        return OPT_ADAMW if dec else OPT_ADAM

    def native(self) -> bool:
        return all(self._native_group(g) for g in self.param_groups)

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:                                # same lazy initialisation as torch/optim/adam.py:_init_group
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _mark_untouched(self, ps, as_none: bool):
        """Remember the state of the gradients this optimiser just zeroed in place.  `as_none`: until a backward pass (or anybody
        else) writes one of them, it counts as ``None`` -- the ``set_to_none=True`` contract without freeing the tensor."""
        for p in ps:
            if as_none and p.grad is not None:
                self._gver[id(p)] = (p.grad, p.grad._version)          # the tensor ITSELF: a fresh gradient that the caching allocator put
            else:                                                      # at the same address with the same version is another object
                self._gver.pop(id(p), None)

    def _has_grad(self, p) -> bool:
        g = p.grad
        if g is None:
            return False
        mark = self._gver.get(id(p))
        return mark is None or mark[0] is not g or mark[1] != g._version

    # ---- torch.optim surface ----
    def zero_grad(self, set_to_none: bool = True):
        """Zero IN PLACE on the device path (one launch): the gradient tensors -- and the pointer table built over them -- stay.
        ``set_to_none=True`` (torch's default, what the reference's ``update()`` relies on): the zeroed gradients count as None for
        the next ``step`` unless a backward pass writes them (see the module docstring)."""
        if not self.native():
            return super().zero_grad(set_to_none)
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            tab = _table(self._tables, f"zero{gi}", [[p.grad for p in ps]], dev)
            _call(CdxOptimArgs(g=tab.row(0), numel=tab.numel_ptr, chunks=tab.chunks.data_ptr(), n_tensors=tab.n_tensors,
                               n_chunks=tab.n_chunks, chunk_elems=CHUNK, mode=OPT_ZERO), dev)
            self._mark_untouched(ps, set_to_none)

    @torch.no_grad()
    def step(self, closure=None, *, max_norm: Optional[float] = None, ema=None, zero_grad: bool = False):
        if not self.native():                           # CPU / unsupported options: the reference's own sequence
            self._sync_steps()
            self._t = {}
            if max_norm:
                self.last_grad_norm = torch.nn.utils.clip_grad_norm_([p for g in self.param_groups for p in g["params"]], max_norm)
            out = super().step(closure)
            if zero_grad:
                super().zero_grad()
            if ema is not None:
                from ..utils.misc import ema_update
                ema_update(ema[0], ema[1], ema[2])
            return out
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # (a gradient this optimiser zeroed in place and nobody wrote since is a `None` gradient: torch skips such a parameter)
        groups = [(g, [p for p in g["params"] if self._has_grad(p)]) for g in self.param_groups]
        groups = [(g, ps) for g, ps in groups if ps]
        if not groups:
            if ema is not None:
                ema_update_native(ema[0], ema[1], ema[2])
            return loss
        dev = groups[0][1][0].device
        if self._norm is None or self._norm.device != dev:
            self._norm = torch.ones(2, dtype=torch.float32, device=dev)
        clip = float(max_norm) if max_norm else 0.0
        if clip > 0.0:
            allp = [p for _, ps in groups for p in ps]
            tab = _table(self._tables, "norm", [[p.grad for p in allp]], dev)
            _call(CdxOptimArgs(g=tab.row(0), numel=tab.numel_ptr, chunks=tab.chunks.data_ptr(), n_tensors=tab.n_tensors,
                               n_chunks=tab.n_chunks, chunk_elems=CHUNK, mode=OPT_SUMSQ, max_norm=clip,
                               partial=tab.partial.data_ptr(), norm=self._norm.data_ptr()), dev)
            self.last_grad_norm = self._norm[0].clone()
        ema_of, rate = None, 0.0
        if ema is not None:
            model, model_ema, rate = ema
            ema_of = {id(p): e for p, e in zip(model.parameters(), model_ema.parameters())}
            if not all(id(p) in ema_of for _, ps in groups for p in ps):
                raise RuntimeError("FusedAdamW: `ema` does not cover the optimiser's parameters")
        for gi, (group, ps_all) in enumerate(groups):
            # one launch per distinct step count (torch keeps the count per parameter: a parameter that joined late, or one whose
            # gradient was None for a while, is bias-corrected with ITS count); normally that is one launch per group
            by_step = {}
            for p in ps_all:
                st = self._state(p)
                t = self._t.get(id(p))
                if t is None:
                    t = int(float(st["step"]))
                self._t[id(p)] = t = t + 1
                by_step.setdefault(t, []).append(p)
            b1, b2 = group["betas"]
            for bi, t in enumerate(sorted(by_step)):        # (tables keyed by the bucket's RANK: the count itself changes every step)
                ps = by_step[t]
                sts = [self.state[p] for p in ps]
                lists = [ps, [p.grad for p in ps], [st["exp_avg"] for st in sts], [st["exp_avg_sq"] for st in sts]]
                if ema_of is not None:
                    lists.append([ema_of[id(p)].detach() for p in ps])
                tab = _table(self._tables, f"adamw{gi}/{bi}", lists, dev)
                _call(CdxOptimArgs(p=tab.row(0), g=tab.row(1), m=tab.row(2), v=tab.row(3), ema=tab.row(4) if ema_of is not None else None,
                                   numel=tab.numel_ptr, chunks=tab.chunks.data_ptr(), n_tensors=tab.n_tensors, n_chunks=tab.n_chunks,
                                   chunk_elems=CHUNK, mode=self._mode_of(group), lr=float(group["lr"]), beta1=float(b1), beta2=float(b2),
                                   eps=float(group["eps"]), weight_decay=float(group["weight_decay"]),
                                   step_size=float(group["lr"]) / (1.0 - b1 ** t), bc2_sqrt=math.sqrt(1.0 - b2 ** t),
                                   ema_rate=float(rate), max_norm=clip, zero_grad=int(zero_grad), norm=self._norm.data_ptr()), dev)
                _bump_versions(ps)
                if ema_of is not None:
                    _bump_versions([ema_of[id(p)] for p in ps])
                if zero_grad:
                    self._mark_untouched(ps, True)
        if ema_of is not None:
            # the reference's ema_update covers EVERY parameter (basic.py:83-86): the ones that took no optimiser step this
            # iteration (no gradient: a condition encoder while update() runs without a condition, an unused branch) in one more launch
            stepped = {id(p) for _, ps in groups for p in ps}
            rest = [(p.detach(), e.detach()) for p, e in zip(ema[0].parameters(), ema[1].parameters()) if id(p) not in stepped]
            if rest:
                if not all(native_device(t) and t.is_contiguous() for pe in rest for t in pe):
                    raise RuntimeError("FusedAdamW: `ema` holds parameters outside the fp32 device path")
                tab = _table(self._tables, "ema_rest", [[p for p, _ in rest], [e for _, e in rest]], dev)
                _call(CdxOptimArgs(p=tab.row(0), ema=tab.row(1), numel=tab.numel_ptr, chunks=tab.chunks.data_ptr(), n_tensors=tab.n_tensors,
                                   n_chunks=tab.n_chunks, chunk_elems=CHUNK, mode=OPT_EMA, ema_rate=float(rate)), dev)
                _bump_versions([e for _, e in rest])
        return loss


class FusedAdamW(_FusedStep, torch.optim.AdamW):
    """``torch.optim.AdamW`` (what ``DiffusionModel`` builds, reference diffusion/basic.py:66) whose ``step`` runs as one multi-tensor
    gfx950 kernel when every parameter is an fp32 tensor on a ROCm device (else: torch's own step)."""
    _OPT_MODE = OPT_ADAMW


class FusedAdam(_FusedStep, torch.optim.Adam):
    """``torch.optim.Adam`` -- L2 weight decay, the optimiser of the reference's classifiers (classifier/base.py:24: lr 2e-4, weight
    decay 1e-4) -- on the same kernel (mode CDX_OPT_ADAM)."""
    _OPT_MODE = OPT_ADAM
