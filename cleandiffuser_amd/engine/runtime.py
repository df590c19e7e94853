"""ctypes binding of libcdx.so (include/cdx.h) + program cache + launch helpers.

PyTorch is plumbing here: it owns device memory and the HIP stream; every tensor crosses the boundary as a raw
device pointer.  There is NO CPU or eager fallback in this module -- if the library cannot be loaded on a ROCm
device, callers get a RuntimeError telling them to build it (``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os
import weakref
from typing import Optional

import numpy as np
import torch

from . import program as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libcdx.so")
LIB_PATH = os.environ.get("CDX_LIB", LIB_PATH)          # A/B hook: run the same process against another build
ABI_VERSION = 8


class CdxStep(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("vsel", ctypes.c_int32), ("noise_idx", ctypes.c_int32),
                ("push", ctypes.c_int32), ("alpha", ctypes.c_float), ("sigma", ctypes.c_float),
                ("k", ctypes.c_float * 5), ("flags", ctypes.c_int32)]


class CdxUnet1dLaunch(ctypes.Structure):
    _fields_ = [
        ("ops", ctypes.c_void_p), ("wblob", ctypes.c_void_p),
        ("n_ops", ctypes.c_int32), ("lds_floats", ctypes.c_int32),
        ("x_off", ctypes.c_int32), ("x_stride", ctypes.c_int32),
        ("pred_off", ctypes.c_int32), ("pred_stride", ctypes.c_int32), ("pred_branch_floats", ctypes.c_int32),
        ("prev_off", ctypes.c_int32), ("scratch_off", ctypes.c_int32),
        ("out_vec_off", ctypes.c_int32), ("out_vec_len", ctypes.c_int32),
        ("tile", ctypes.c_int32), ("cond_slot_off", ctypes.c_int32), ("cond_slot_stride", ctypes.c_int32),
        ("cond_coff", ctypes.c_int32), ("cond_dim", ctypes.c_int32),
        ("zero_off", ctypes.c_int32), ("zero_floats", ctypes.c_int32), ("zrow_off", ctypes.c_int32),
        ("prof_off", ctypes.c_int32), ("items_in_lds", ctypes.c_int32), ("desc_off", ctypes.c_int32),
        ("desc_words", ctypes.c_int32),
        ("batch", ctypes.c_int32), ("horizon", ctypes.c_int32), ("dim", ctypes.c_int32), ("emb_dim", ctypes.c_int32),
        ("temb", ctypes.c_void_p), ("steps", ctypes.c_void_p),
        ("n_steps", ctypes.c_int32), ("temb_per_sample", ctypes.c_int32), ("predict_noise", ctypes.c_int32),
        ("cfg_mode", ctypes.c_int32), ("cfg_w", ctypes.c_float),
        ("cond", ctypes.c_void_p), ("x_in", ctypes.c_void_p), ("prior", ctypes.c_void_p),
        ("fix_mask", ctypes.c_void_p), ("noise", ctypes.c_void_p), ("x_min", ctypes.c_void_p),
        ("x_max", ctypes.c_void_p), ("x_out", ctypes.c_void_p), ("prof", ctypes.c_void_p)]


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
    lib.cdx_unet1d_run.argtypes = [ctypes.POINTER(CdxUnet1dLaunch), ctypes.c_void_p]
    lib.cdx_unet1d_run.restype = ctypes.c_int
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


# ------------------------------------------------------------------------------------------------ #
# program cache: one compiled program per (module, horizon), invalidated when any parameter changes #
# ------------------------------------------------------------------------------------------------ #
class _Compiled:
    def __init__(self, prog: P.Program, sig):
        self.prog = prog
        self.sig = sig
        self.ops_dev = torch.from_numpy(prog.ops_buffer.copy()).to(prog.blob.device)


_cache = weakref.WeakKeyDictionary()


def _signature(module):
    """Identity of a module's weights: storage pointers + autograd version counters + the explicit epoch that
    ``utils.invalidate_weights`` / ``ema_update`` / ``load`` bump (``p.data`` writes leave ``_version`` untouched)."""
    return (module.__dict__.get("_cdx_epoch", 0),) + tuple((p.data_ptr(), p._version) for p in module.parameters()) + \
        tuple((b.data_ptr(), b._version) for b in module.buffers())


def compiled_program(module, horizon: int, edm: bool = False) -> _Compiled:
    """The module's program at this horizon.  `edm`: with the two extra dense state buffers of the EDM / consistency step kinds
    (2 x H x D floats of LDS more -- the difference between fitting and not fitting for the shipped Diffuser kitchen net)."""
    per_mod = _cache.setdefault(module, {})
    sig = _signature(module)
    key = (horizon, bool(edm) and not _is_half_janner(module))
    hit = per_mod.get(key)
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        kind = _mlp_kind(module)
        if kind == "pearce":
            prog = P.compile_pearce_mlp(module, horizon, edm=edm)
        elif kind == "dql":
            prog = P.compile_dql_mlp(module, horizon, edm=edm)
        elif kind == "sfbc":
            prog = P.compile_sfbc_unet(module, horizon, edm=edm)
        elif kind == "mlpnn":
            prog = P.compile_mlp_nn(module, horizon, edm=edm)
        elif _is_chiunet(module):
            prog = P.compile_chiunet(module, horizon, edm=edm)
        elif _is_half_janner(module):
            prog = P.compile_half_janner(module, horizon)
        else:
            prog = P.compile_janner(module, horizon, edm=edm)
    per_mod[key] = _Compiled(prog, sig)
    return per_mod[key]


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
        if lins and all(l.out_features % 16 == 0 and l.out_features <= 1024 for l in lins) and all(P._act_id(a) is not None for a in acts):
            return "mlpnn"
    return None


def _is_chiunet(module) -> bool:
    from ..nn_diffusion.chiunet import ChiUNet1d
    return type(module) is ChiUNet1d


def _is_half_janner(module) -> bool:
    from ..nn_classifier.half_jannerunet import HalfJannerUNet1d
    return isinstance(module, HalfJannerUNet1d)


def supported_backbone(module, horizon: int, edm: bool = False) -> Optional[str]:
    """None if the fused kernel can run `module` at this horizon, else a human-readable reason."""
    if _is_half_janner(module):
        why = P.supports_half_janner(module)
        if why:
            return why
        if horizon != module.horizon:
            return f"classifier was built for horizon {module.horizon}"
        try:
            compiled_program(module, horizon)
        except ValueError as e:
            return str(e)
        return None
    if _is_chiunet(module):
        why = P.supports_chiunet(module)
        if why:
            return why
        n_down = sum(1 for lvl in module.downs if not isinstance(lvl[2], torch.nn.Identity))
        if horizon % (1 << n_down) != 0:
            return f"horizon {horizon} not divisible by 2^{n_down}"
        try:
            compiled_program(module, horizon, edm)
        except ValueError as e:                      # LDS plan does not fit one workgroup
            return str(e)
        return None
    if not _is_janner(module):
        return f"{type(module).__name__} has no fused program yet"
    why = P.supports_janner(module)
    if why:
        return why
    n_down = sum(1 for lvl in module.downs if not isinstance(lvl[3], torch.nn.Identity))
    if horizon % (1 << n_down) != 0:
        return f"horizon {horizon} not divisible by 2^{n_down}"
    try:
        compiled_program(module, horizon, edm)           # cached: the caller's own compiled_program() call is a hit
    except ValueError as e:                              # wide / long configurations whose LDS plan exceeds one workgroup
        return str(e)
    return None


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
_timing = {"on": False, "events": []}


def enable_launch_timing(on: bool):
    _timing["on"] = bool(on)
    _timing["events"] = []


def drain_launch_timing():
    """Synchronise and return the duration (ms) of every launch recorded since enable_launch_timing(True)."""
    out = []
    for start, end in _timing["events"]:
        end.synchronize()
        out.append(start.elapsed_time(end))
    _timing["events"] = []
    return out


_prof = {"buf": None}


def set_profile_buffer(buf: Optional[torch.Tensor]):
    """int64 device tensor of n_ops*4+2 entries (or None): workgroup 0 stamps s_memtime per op (debug aid)."""
    _prof["buf"] = buf


def _launch(comp: _Compiled, **kw):
    if kw["batch"] <= 0:                       # empty request: nothing to launch, outputs are already empty tensors
        return
    _launch_nonempty(comp, **kw)


def describe_launch(comp: _Compiled, *, batch, x_in=None, x_out=None, temb=None, steps_dev=None, n_steps=0, temb_per_sample=0,
                    predict_noise=0, cfg_mode=0, cfg_w=0.0, cond=None, prior=None, fix_mask=None, noise=None,
                    x_min=None, x_max=None) -> CdxUnet1dLaunch:
    """The cdx_unet1d_launch block of a request (tensors may be filled in later by a caller that sequences several launches)."""
    prog = comp.prog
    return CdxUnet1dLaunch(
        ops=comp.ops_dev.data_ptr(), wblob=prog.blob.data_ptr(), n_ops=len(prog.ops), lds_floats=prog.lds_floats,
        x_off=prog.x_off, x_stride=prog.x_stride, pred_off=prog.pred_off, pred_stride=prog.pred_stride,
        pred_branch_floats=prog.pred_branch_floats, prev_off=prog.prev_off, scratch_off=prog.scratch_off,
        out_vec_off=prog.out_vec_off, out_vec_len=prog.out_vec_len,
        tile=prog.tile, cond_slot_off=prog.cond_slot_off, cond_slot_stride=prog.cond_slot_stride,
        cond_coff=prog.cond_coff, cond_dim=prog.cond_dim, zero_off=prog.zero_off, zero_floats=prog.zero_floats, zrow_off=prog.zrow_off,
        prof_off=prog.prof_off, items_in_lds=int(prog.items_in_lds), desc_off=prog.desc_off,
        desc_words=prog.desc_words,
        batch=batch, horizon=prog.horizon, dim=prog.dim, emb_dim=prog.emb_dim,
        temb=_ptr(temb), steps=_ptr(steps_dev), n_steps=n_steps, temb_per_sample=temb_per_sample,
        predict_noise=int(predict_noise), cfg_mode=cfg_mode, cfg_w=float(cfg_w), cond=_ptr(cond),
        x_in=_ptr(x_in), prior=_ptr(prior), fix_mask=_ptr(fix_mask), noise=_ptr(noise),
        x_min=_ptr(x_min), x_max=_ptr(x_max), x_out=_ptr(x_out), prof=_ptr(_prof["buf"]))


def _launch_nonempty(comp: _Compiled, *, x_in, **kw):
    L = describe_launch(comp, x_in=x_in, **kw)
    if _timing["on"]:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(torch.cuda.current_stream(x_in.device))
    _check(load_library().cdx_unet1d_run(ctypes.byref(L), _stream_ptr(x_in.device)), "cdx_unet1d_run")
    if _timing["on"]:
        end.record(torch.cuda.current_stream(x_in.device))
        _timing["events"].append((start, end))


# ------------------------------------------------------------------------------------------------ #
# entry points used by dispatch.py                                                                   #
# ------------------------------------------------------------------------------------------------ #
def _backbone_cond(module, prog, condition, device):
    """Condition tensor in the layout the program expects, None for "no condition", False for "cannot be fused"."""
    if _is_chiunet(module):
        if condition is None:
            return False                              # the reference raises on a missing condition (Q12): keep that path
        c = _f32c(torch.flatten(condition, 1), device)
        return c if c.shape[1] == prog.cond_dim else False
    if condition is None:
        return None
    if condition.dim() != 2 or condition.shape[1] != prog.emb_dim:
        return False
    return _f32c(condition, device)


def backbone_forward(module, x, noise, condition) -> Optional[torch.Tensor]:
    """``BaseNNDiffusion.forward`` on the device: one launch, per-sample timesteps."""
    if x.dim() == 3 and (_is_janner(module) or _is_chiunet(module)):
        from . import runtime2                        # second-generation kernel: one FiLM row per sample
        y = runtime2.backbone_forward2(module, x, noise, condition)
        if y is not None:
            return y
    if x.dim() == 3 and _is_half_janner(module) and condition is None and module.out_dim == 1:
        from . import runtime2                        # the classifier's own v2 program: log p of the batch in one launch
        y = runtime2.classifier_forward2(module, x, noise)
        if y is not None:
            return y
    if x.dim() != 3 or supported_backbone(module, x.shape[1]) is not None:
        return None
    load_library()
    b, h, d = x.shape
    with torch.no_grad():
        comp = compiled_program(module, h)
        temb = _f32c(module.map_noise(noise), x.device)
        cond = _backbone_cond(module, comp.prog, condition, x.device)
        if cond is False:
            return None
        xin = _f32c(x, x.device)
        vec_len = comp.prog.out_vec_len
        out = torch.empty((b, vec_len), device=x.device, dtype=torch.float32) if vec_len else torch.empty_like(xin)
        _launch(comp, batch=b, x_in=xin, x_out=out, temb=temb, temb_per_sample=1,
                cfg_mode=1 if cond is not None else 0, cond=cond)
    return out


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


def fused_sample_mlp(solver, net, kind, plan, xt, prior, cond_vec, w_cfg, feed) -> Optional[torch.Tensor]:
    """Batch-tiled MLP denoisers (x of shape (B, D)): one workgroup per `MLP_TILE` samples, whole loop in one launch."""
    b, d = xt.shape
    dev = xt.device
    tile = mlp_tile(b)
    n_tiles = -(-b // tile)
    pad = n_tiles * tile - b
    try:
        fix_mask = _dense_hd(solver.fix_mask, 1, d, dev)
        clip = getattr(plan, "clip_each_step", True)
        x_min = _dense_hd(getattr(solver, "x_min", None), 1, d, dev) if clip else None
        x_max = _dense_hd(getattr(solver, "x_max", None), 1, d, dev) if clip else None
    except (ValueError, RuntimeError):
        return None

    def rows(t):                                      # (B, D) -> (n_tiles * tile, D), zero rows appended
        t = _f32c(t, dev)
        return torch.cat([t, t.new_zeros(pad, *t.shape[1:])]) if pad else t

    def table(t):                                     # (1, D) -> (tile, D): the kernel indexes bounds/masks per tile row
        return None if t is None else t.expand(tile, d).contiguous()

    cond = None
    if cond_vec is None and w_cfg not in (0.0, 1.0):
        return None                                   # the reference raises here; let the torch executor do it
    if cond_vec is not None and w_cfg != 0.0:         # w = 1: one conditional forward; otherwise the cond | zeros pair per step
        cond = rows(torch.flatten(cond_vec, 1))
    load_library()
    with torch.no_grad():
        try:
            comp = compiled_program(net, tile, plan_is_edm(plan))
        except ValueError:                            # very wide nets: the tile's LDS plan exceeds one workgroup -> PyTorch executor
            return None
        if cond is not None and cond.shape[1] != comp.prog.cond_dim:
            return None
        t_vec = device_times(plan, dev)
        temb = _f32c(net.map_noise(t_vec), dev)
        if kind == "sfbc":                            # SfBCUNet: the batch-invariant t_layer runs here, once per step record
            temb = _f32c(net.t_layer(temb), dev)
        if kind == "pearce":                          # PearceMlp also consumes the raw timestep as a feature (Q11)
            temb = torch.cat([temb, t_vec.to(torch.float32).unsqueeze(1)], 1).contiguous()
        steps_dev = steps_to_device(plan, dev)
        noise = feed.many(xt, plan.n_noise)
        if noise is not None and pad:                 # zero rows behind every draw: the last tile's unused samples
            noise = torch.cat([_f32c(noise, dev), noise.new_zeros(noise.shape[0], pad, d)], dim=1).contiguous()
        xin = rows(xt)
        out = torch.empty_like(xin)
        _launch(comp, batch=n_tiles, x_in=xin, x_out=out, temb=temb, steps_dev=steps_dev, n_steps=len(plan.steps),
                predict_noise=_predicts_noise(plan, solver), cfg_mode=(0 if cond is None else (1 if w_cfg == 1.0 else 2)), cfg_w=w_cfg,
                cond=cond,
                prior=rows(prior) if fix_mask is not None else None, fix_mask=table(fix_mask), noise=noise,
                x_min=table(x_min), x_max=table(x_max))
    return out[:b]


def fused_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, feed, x_scale: Optional[float] = None) -> Optional[torch.Tensor]:
    """Whole denoising loop in one launch.  Returns None when this request must take the PyTorch executor.
    `x_scale` given: `xt` is the raw N(0, I) draw (see dispatch.try_fused_raw); only the v2 U-Net kernel takes such a request."""
    net = model["diffusion"]
    if xt.dim() == 2 and _mlp_kind(net) is not None:
        from . import runtime2                        # the second-generation kernel first (it returns None BEFORE drawing from `feed`)
        load_library()
        out = runtime2.fused_sample_mlp2(solver, net, _mlp_kind(net), plan, xt, prior, cond_vec, w_cfg, feed)
        if out is not None:
            return out
        return fused_sample_mlp(solver, net, _mlp_kind(net), plan, xt, prior, cond_vec, w_cfg, feed)
    if xt.dim() != 3 or not (_is_janner(net) or _is_chiunet(net)):
        return None
    v1_why = supported_backbone(net, xt.shape[1], plan_is_edm(plan))     # (nets too large for the first kernel's LDS plan may
    #                                                                       still fit the second one's compact program)
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
    if _is_janner(net) or _is_chiunet(net):
        from . import runtime2                        # the second-generation kernel (conditional / CFG / EDM included)
        out = runtime2.fused_sample2(solver, net, plan, xt, prior, feed, fix_mask, x_min, x_max, x_scale=x_scale,
                                     cond=cond_vec if w_cfg != 0.0 else None, w_cfg=w_cfg)
        if out is not None:
            return out
    if x_scale is not None or v1_why is not None:
        return None                                   # raw-draw requests are only taken by the v2 kernel; the caller forms x_T itself
    with torch.no_grad():
        comp = compiled_program(net, h, plan_is_edm(plan))
        # every eligibility check that can still send the request to the PyTorch executor comes BEFORE the first draw from `feed`:
        # a recorded noise list must reach that executor unconsumed
        if cond_vec is None or w_cfg == 0.0:
            mode, cond = 0, None
        else:
            mode, cond = (1 if w_cfg == 1.0 else 2), _backbone_cond(net, comp.prog, cond_vec, dev)
            if cond is False or cond is None:
                return None
        t_vec = device_times(plan, dev)
        temb = _f32c(net.map_noise(t_vec), dev)
        steps_dev = steps_to_device(plan, dev)
        noise = feed.many(xt, plan.n_noise)
        xin = _f32c(xt, dev)
        out = torch.empty_like(xin)
        _launch(comp, batch=b, x_in=xin, x_out=out, temb=temb, steps_dev=steps_dev, n_steps=len(plan.steps),
                predict_noise=_predicts_noise(plan, solver), cfg_mode=mode, cfg_w=w_cfg, cond=cond,
                prior=_f32c(prior, dev) if fix_mask is not None else None, fix_mask=fix_mask, noise=noise,
                x_min=x_min, x_max=x_max)
    return out


def probe_mfma_layout(device="cuda:0") -> torch.Tensor:
    out = torch.zeros(4, 64, 4, device=device)
    _check(load_library().cdx_probe_mfma_layout(out.data_ptr(), _stream_ptr(out.device)), "cdx_probe_mfma_layout")
    torch.cuda.synchronize(out.device)
    return out.cpu()
