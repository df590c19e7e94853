"""ctypes binding of libcdx.so (include/cdx.h): library loading, step records, request routing into the program kernel.

PyTorch is plumbing here: it owns device memory and the HIP stream; every tensor crosses the boundary as a raw
device pointer.  There is NO CPU or eager fallback in this module -- if the library cannot be loaded on a ROCm
device, callers get a RuntimeError telling them to build it (``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from .consts import _act_id


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libcdx.so")
LIB_PATH = os.environ.get("CDX_LIB", LIB_PATH)          # A/B hook: run the same process against another build
ABI_VERSION = 17


class CdxStep(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("vsel", ctypes.c_int32), ("noise_idx", ctypes.c_int32),
                ("push", ctypes.c_int32), ("alpha", ctypes.c_float), ("sigma", ctypes.c_float),
                ("k", ctypes.c_float * 5), ("flags", ctypes.c_int32)]


_lib = None


def load_library(path: Optional[str] = None):
    """Load libcdx.so and declare prototypes.  Raises RuntimeError (never falls back) if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"cleandiffuser_amd: native library not found at {path}.  Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).  "
            f"There is no eager fallback for supported backbones on a ROCm device.")
    lib = ctypes.CDLL(path)
    lib.cdx_abi_version.restype = ctypes.c_int
    lib.cdx_last_error.restype = ctypes.c_char_p
    lib.cdx_probe_mfma_layout.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cdx_probe_mfma_layout.restype = ctypes.c_int
    if lib.cdx_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libcdx.so ABI {lib.cdx_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().cdx_last_error().decode()
        raise RuntimeError(f"{what} failed with code {rc}: {msg}")


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _Flat:
    """Where a module tree keeps its tensors: [(owner's _parameters / _buffers dict, name)] plus what it takes to notice that the tree
    itself changed ([(parent, child name, child, number of children / tensors it had)]).  ``module.parameters()`` walks the tree through
    three layers of Python generators -- 0.27 ms for the config-2 U-Net, six times per ``sample()`` call: 1.6 of the 1.7 ms of host time
    of a steady-state call (tools/host_profile.py, round 4)."""

    def __init__(self, module):
        self.slots, self.tree = [], []
        seen = set()
        for mod in module.modules():
            self.tree.append((mod, len(mod._modules), self._live(mod._parameters), self._live(mod._buffers), list(mod._modules.items())))
            for d in (mod._parameters, mod._buffers):
                for name, t in d.items():
                    if t is not None and id(t) not in seen:          # (shared tensors once, like module.parameters())
                        seen.add(id(t))
                        self.slots.append((d, name))

    @staticmethod
    def _live(d) -> int:
        """how many slots of a _parameters / _buffers dict hold a tensor: a slot that goes from None to a tensor (register_buffer(name,
        None) filled in later) keeps the dict's length but must enter the signature"""
        n = 0
        for t in d.values():
            n += t is not None
        return n

    def valid(self) -> bool:
        for mod, n_mod, n_par, n_buf, children in self.tree:
            if len(mod._modules) != n_mod or self._live(mod._parameters) != n_par or self._live(mod._buffers) != n_buf:
                return False
            for name, child in children:
                if mod._modules.get(name) is not child:
                    return False
        return True


class _Scope(__import__("threading").local):
    """per thread: a second thread entering a scope must not reuse the id under which the first one cached signatures"""
    depth = 0
    id = 0


_sig_scope = _Scope()
_scope_ids = __import__("itertools").count(1)             # process-wide: no two scopes, on whatever thread, share an id


class signature_scope:
    """``with signature_scope():`` around ONE request (a sample() call, a backbone forward): weights cannot change while it runs, so every
    module's signature is computed once inside it however many caches ask (six times per steady-state sample() call otherwise: 0.33 of
    0.39 ms of host time, tools/host_profile.py).  Outside a scope every call computes afresh."""

    def __enter__(self):
        if _sig_scope.depth == 0:
            _sig_scope.id = next(_scope_ids)
        _sig_scope.depth += 1

    def __exit__(self, *exc):
        _sig_scope.depth -= 1
        return False


def _signature(module):
    if _sig_scope.depth > 0:
        hit = module.__dict__.get("_cdx_sig")
        if hit is not None and hit[0] == _sig_scope.id:
            return hit[1]
        sig = _signature_now(module)
        module.__dict__["_cdx_sig"] = (_sig_scope.id, sig)
        return sig
    return _signature_now(module)


def _signature_now(module):
    """Identity of a module's weights: storage pointers + autograd version counters + the explicit epoch that
    ``utils.invalidate_weights`` / ``ema_update`` / ``load`` bump (``p.data`` writes leave ``_version`` untouched).  Tensors are looked
    up live in their owners' dicts (a replaced Parameter is seen); the list of owners is rebuilt when a submodule was replaced, added or
    removed."""
    flat = module.__dict__.get("_cdx_flat")
    if flat is None or not flat.valid():
        flat = module.__dict__["_cdx_flat"] = _Flat(module)
    sig = [module.__dict__.get("_cdx_epoch", 0)]
    for d, name in flat.slots:
        t = d.get(name)
        if t is None:                      # a tensor was deleted or set to None: describe the tree afresh
            flat = module.__dict__["_cdx_flat"] = _Flat(module)
            return (module.__dict__.get("_cdx_epoch", 0),) + tuple((d2[n2].data_ptr(), d2[n2]._version) for d2, n2 in flat.slots)
        sig.append((t.data_ptr(), t._version))
    return tuple(sig)


def plan_is_edm(plan) -> bool:
    return any(st.kind >= 5 for st in plan.steps)


def _is_janner(module) -> bool:
    from ..nn_diffusion.jannerunet import JannerUNet1d
    return isinstance(module, JannerUNet1d)


def _mlp_kind(module) -> Optional[str]:
    """Batch-tiled MLP programs the compiler knows: 'pearce' | 'dql' | None."""
    from ..nn_diffusion.mlp_backbones import DQLMlp, DVInvMlp, PearceMlp
    if type(module) is PearceMlp and module.hidden_dim % 64 == 0 and module.hidden_dim <= 1024:
        return "pearce"
    if type(module) is DQLMlp or (type(module) is DVInvMlp and module.mid_layer[0].out_features % 64 == 0
                                  and module.mid_layer[0].out_features <= 1024):
        return "dql"
    from ..nn_diffusion.sfbc_unet import SfBCUNet
    if type(module) is SfBCUNet and all(blk.linear1[0].out_features % 16 == 0 and blk.linear1[0].out_features <= 1024
                                        for blk in list(module.down_blocks) + [module.mid_block] + list(module.up_blocks)):
        return "sfbc"
    from ..nn_diffusion.mlp_backbones import MlpNNDiffusion
    if type(module) is MlpNNDiffusion:
        lins = [m[0] for m in module.mlp.mlp if isinstance(m, torch.nn.Sequential)]
        acts = [m[1] for m in module.mlp.mlp if isinstance(m, torch.nn.Sequential)] + [module.mlp.mlp[-1]]
        if lins and all(l.out_features % 16 == 0 and l.out_features <= 1024 for l in lins) and all(_act_id(a) is not None for a in acts):
            return "mlpnn"
    return None


def _is_chiunet(module) -> bool:
    from ..nn_diffusion.chiunet import ChiUNet1d
    return type(module) is ChiUNet1d


def _is_half_janner(module) -> bool:
    from ..nn_classifier.half_jannerunet import HalfJannerUNet1d
    return isinstance(module, HalfJannerUNet1d)


def supported_backbone(module, horizon: int, edm: bool = False) -> Optional[str]:
    """None if the program kernel can run `module` (a temporal U-Net or the HalfJannerUNet1d classifier) at this horizon, else a
    human-readable reason.  (`edm`: kept for callers; EDM plans keep their state in the launch workspace, not in LDS.)"""
    from . import runtime2
    if _is_half_janner(module):
        if horizon != module.horizon:
            return f"classifier was built for horizon {module.horizon}"
        return runtime2.compiled_classifier2(module, horizon).why
    return runtime2.supported(module, horizon)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def _dense_hd(v, h: int, d: int, device) -> Optional[torch.Tensor]:
    """Broadcast a reference-style (1,H,D)/(H,D)/(D,)/scalar tensor to a dense (H, D) fp32 table; None stays None."""
    if v is None or (not isinstance(v, torch.Tensor) and v == 0):
        return None
    memo = None
    if isinstance(v, torch.Tensor):
        # a (1, 1, D) bound would otherwise cost an expand + copy launch in every sample() call; the entry pins its source tensor,
        # so (data_ptr, _version) cannot be recycled under it
        memo = (v.data_ptr(), v._version, tuple(v.shape), str(v.dtype), h, d, str(device))
        hit = _dense_memo.get(memo)
        if hit is not None:
            return hit[1]
    t = torch.as_tensor(v, dtype=torch.float32, device=device)
    if t.dim() >= 3:
        if t.shape[0] != 1:
            raise ValueError("per-sample bounds/masks are not supported by the fused executor")
        t = t[0]
    out = t.expand(h, d).contiguous()
    if memo is not None:
        if len(_dense_memo) >= 64:
            _dense_memo.clear()
        _dense_memo[memo] = (v, out)
    return out


_dense_memo = {}


# ------------------------------------------------------------------------------------------------ #
# optional per-launch timing (HIP events on the launch stream) -- used by bench.py's roofline leg      #
# ------------------------------------------------------------------------------------------------ #
_timing = {"on": False, "events": [], "repair_events": []}


def enable_launch_timing(on: bool):
    _timing["on"] = bool(on)
    _timing["events"] = []
    _timing["repair_events"] = []


def drain_launch_timing():
    """Synchronise and return the duration (ms) of every launch recorded since enable_launch_timing(True)."""
    out = []
    for start, end in _timing["events"]:
        end.synchronize()
        out.append(start.elapsed_time(end))
    _timing["events"] = []
    return out


def drain_repair_timing():
    """The same for the REPAIR launches behind split / grouped launches (cdx.h: run_if) -- recorded apart: a repair launch that finds no
    error is an empty grid, and averaging it into the kernel time would halve it."""
    out = []
    for start, end in _timing["repair_events"]:
        end.synchronize()
        out.append(start.elapsed_time(end))
    _timing["repair_events"] = []
    return out


_prof = {"buf": None}


def set_profile_buffer(buf: Optional[torch.Tensor]):
    """int64 device tensor of n_ops*4+2 entries (or None): workgroup 0 stamps s_memtime per op (debug aid)."""
    _prof["buf"] = buf


def backbone_forward(module, x, noise, condition) -> Optional[torch.Tensor]:
    """``BaseNNDiffusion.forward`` / the classifier's forward on the device: one launch of the program kernel, per-sample timesteps
    (one FiLM row per sample).  None -> the caller keeps the PyTorch modules."""
    if x.dim() != 3:
        return None
    from . import runtime2
    load_library()
    if _is_janner(module) or _is_chiunet(module):
        return runtime2.backbone_forward2(module, x, noise, condition)
    if _is_half_janner(module) and condition is None and module.out_dim == 1:
        return runtime2.classifier_forward2(module, x, noise)       # log p of the batch in one launch
    return None


def _predicts_noise(plan, solver) -> bool:
    """What the network output means for this plan: the plan's own statement (rectified flow) or the solver's attribute."""
    own = getattr(plan, "network_predicts_noise", None)
    return bool(getattr(solver, "predict_noise", False) if own is None else own)


def device_times(plan, device) -> torch.Tensor:
    """The timestep of every step record as one device vector (int64 grid indices or fp32 times), memoised on the plan."""
    from .plan import cached
    dt = torch.long if plan.t_is_integer else torch.float32
    return cached(plan, ("t", str(device)), lambda: torch.tensor([st.t for st in plan.steps], dtype=dt, device=device))


def steps_to_device(plan, device) -> torch.Tensor:
    from .plan import cached
    return cached(plan, ("steps", str(device)), lambda: _pack_steps(plan, device))


def _pack_steps(plan, device) -> torch.Tensor:
    arr = (CdxStep * len(plan.steps))()
    k = 0
    for i, st in enumerate(plan.steps):
        arr[i].kind, arr[i].vsel, arr[i].push, arr[i].flags = st.kind, st.vsel, int(st.push), int(st.flags)
        arr[i].alpha, arr[i].sigma = st.alpha, st.sigma
        for j in range(5):
            arr[i].k[j] = st.k[j]
        if st.noise:
            arr[i].noise_idx, k = k, k + 1
        else:
            arr[i].noise_idx = -1
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device)


def mlp_tile(batch: int) -> int:
    """Samples per workgroup of a batch-tiled MLP program: 16 (one 16x16x4 MFMA column tile) once that still gives every CU a
    workgroup, else 8 or 4 (4x4x1 MFMA column blocks) -- BASELINE config 1 is B = 256: 16 workgroups of 16 samples leave 240 of the
    256 CUs idle, 64 workgroups of 4 do a quarter of the work each.  CDX_MLP_TILE forces 4, 8 or 16."""
    forced = os.environ.get("CDX_MLP_TILE")
    if forced in ("4", "8", "16"):
        return int(forced)
    for tile in (16, 8):
        if batch >= 256 * tile:
            return tile
    return 4 if batch <= 1024 else 8


def fused_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, feed, x_scale: Optional[float] = None) -> Optional[torch.Tensor]:
    """Whole denoising loop in one launch of the program kernel.  Returns None when this request must take another executor (every
    None comes BEFORE the first draw from `feed`: a recorded noise list must reach that executor unconsumed).
    `x_scale` given: `xt` is the raw N(0, I) draw (see dispatch.try_fused_raw) and the kernel forms x_T itself."""
    net = model["diffusion"]
    from . import runtime2
    if xt.dim() == 2 and _mlp_kind(net) is not None:
        load_library()
        return runtime2.fused_sample_mlp2(solver, net, _mlp_kind(net), plan, xt, prior, cond_vec, w_cfg, feed)
    if xt.dim() != 3 or not (_is_janner(net) or _is_chiunet(net)):
        return None
    if cond_vec is None and w_cfg not in (0.0, 1.0):
        return None                                   # the reference raises here; let the torch executor do it
    if _is_chiunet(net) and (cond_vec is None or w_cfg == 0.0):
        return None                                   # ChiUNet1d cannot run unconditionally (reference raises)
    try:
        b, h, d = xt.shape
        dev = xt.device
        fix_mask = _dense_hd(solver.fix_mask, h, d, dev)
        clip = getattr(plan, "clip_each_step", True)
        x_min = _dense_hd(getattr(solver, "x_min", None), h, d, dev) if clip else None
        x_max = _dense_hd(getattr(solver, "x_max", None), h, d, dev) if clip else None
    except ValueError:
        return None
    load_library()
    return runtime2.fused_sample2(solver, net, plan, xt, prior, feed, fix_mask, x_min, x_max, x_scale=x_scale,
                                  cond=cond_vec if w_cfg != 0.0 else None, w_cfg=w_cfg)


def probe_mfma_layout(device="cuda:0") -> torch.Tensor:
    out = torch.zeros(4, 64, 4, device=device)
    _check(load_library().cdx_probe_mfma_layout(out.data_ptr(), _stream_ptr(out.device)), "cdx_probe_mfma_layout")
    torch.cuda.synchronize(out.device)
    return out.cpu()
