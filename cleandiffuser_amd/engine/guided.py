"""Classifier-guided sampling loop in one native call (``cdx_guided_run``, include/cdx.h).

Reference behaviour: ``DiscreteDiffusionSDE.sample`` / ``ContinuousDiffusionSDE.sample`` with ``w_cg > 0``
(diffusionsde.py:526-594 + classifier_guidance :153-173) -- per step one backbone forward, one ``classifier.gradients`` call
(autograd in the reference) and the solver update.  Here every step is: the program kernel in forward mode (or the implicit-GEMM
U-Net executor for nets that do not fit one workgroup), the explicit
classifier forward+backward (engine/classifier_grad.py) and the solver-step kernel with the guidance shift folded in; all
launches of all steps are enqueued by a single C call.
"""
import ctypes
from typing import Optional

import numpy as np
import torch

from . import classifier_grad, runtime
from .bigbatch import host_steps
from .runtime import CdxStep, _check, _dense_hd, _f32c, _predicts_noise, _stream_ptr, load_library
from .runtime2 import CdxUnet2Launch

_FP, _I = ctypes.c_void_p, ctypes.c_int32


class CdxGuidedLaunch(ctypes.Structure):
    _fields_ = [("denoiser", ctypes.POINTER(CdxUnet2Launch)), ("classifier", ctypes.POINTER(classifier_grad.CdxHjgradWeights)),
                ("steps", ctypes.POINTER(CdxStep)), ("cg_scale", ctypes.POINTER(ctypes.c_float)), ("n_steps", _I), ("batch", _I),
                ("hd", _I), ("predict_noise", _I), ("temb", _FP), ("clf_emb0", _FP), ("x_in", _FP), ("prior", _FP),
                ("fix_mask", _FP), ("noise", _FP), ("x_min", _FP), ("x_max", _FP), ("x_out", _FP), ("workspace", _FP),
                ("workspace_floats", ctypes.c_longlong), ("denoiser_gemm", _FP), ("denoiser_emb_dim", _I), ("denoiser_chunk", _I)]


_declared = False
_ws = {}


def _lib():
    global _declared
    lib = load_library()
    if not _declared:
        lib.cdx_guided_workspace_floats.argtypes = [ctypes.POINTER(CdxGuidedLaunch)]
        lib.cdx_guided_workspace_floats.restype = ctypes.c_longlong
        lib.cdx_guided_run.argtypes = [ctypes.POINTER(CdxGuidedLaunch), ctypes.c_void_p]
        lib.cdx_guided_run.restype = ctypes.c_int
        _declared = True
    return lib


def guided_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, feed) -> Optional[torch.Tensor]:
    """None -> the caller steps the loop itself (PyTorch executor, which still uses the native pieces per step)."""
    from ..classifier.rew_classifiers import CumRewClassifier
    from ..nn_classifier.half_jannerunet import HalfJannerUNet1d
    from ..diffusion.diffusionsde import BaseDiffusionSDE
    from ..diffusion.newedm import ContinuousEDM
    net, clf = model["diffusion"], solver.classifier
    edm = type(solver) is ContinuousEDM and runtime.plan_is_edm(plan)
    if not (isinstance(solver, BaseDiffusionSDE) or edm):        # the legacy classes shift the prediction after their own conversions
        return None
    if xt.dim() != 3 or type(clf) is not CumRewClassifier or type(clf.model_ema) is not HalfJannerUNet1d:
        return None
    b, h, d = xt.shape
    if not (runtime._is_janner(net) or runtime._is_chiunet(net)):
        return None
    from . import runtime2
    use_cond = cond_vec is not None and w_cfg != 0.0
    gemm_bound, den_comp = None, None
    if runtime2.supported(net, h) is None and not runtime2.compact_only(net, h):
        den_comp, den_t = runtime2.shape_for(net, h, b)              # the program kernel in forward mode, one launch per step
    else:
        # LDS plan larger than one workgroup (the shipped antmaze Diffuser): the per-step denoiser forward runs on the implicit-GEMM
        # U-Net executor inside the same cdx_guided_run call -- unconditional JannerUNet1d only
        from . import bigbatch
        if not runtime._is_janner(net) or use_cond:
            return None
        gemm_bound = bigbatch._bound(net, ("chiunet", h), lambda: bigbatch._bind_janner_gemm(net, h, xt.device))
        if gemm_bound is None or d != gemm_bound.struct.act_dim:
            return None
    if runtime._is_chiunet(net) and not use_cond:
        return None                                                  # ChiUNet1d cannot run unconditionally (the reference raises)
    if edm:
        if any(st.kind not in (5, 6) for st in plan.steps):              # (consistency records have no guided form)
            return None
    elif any(st.kind > 2 for st in plan.steps):
        return None
    if cond_vec is not None and w_cfg not in (0.0, 1.0):
        return None
    if clf.model_ema.horizon != h or clf.model_ema.in_dim != d:
        return None
    dev = xt.device
    bound = classifier_grad.bound_for(clf.model_ema, dev)
    if bound is None:
        return None
    try:
        clip = getattr(plan, "clip_each_step", True)
        fix_mask = _dense_hd(solver.fix_mask, h, d, dev)
        x_min = _dense_hd(getattr(solver, "x_min", None), h, d, dev) if clip else None
        x_max = _dense_hd(getattr(solver, "x_max", None), h, d, dev) if clip else None
    except ValueError:
        return None
    if not use_cond and runtime._is_janner(net) and not edm:
        # unconditional temporal U-Net: the classifier's forward + backward joins the denoiser in the program kernel -- the whole
        # guided loop is one launch instead of ~105 launches per step
        out = runtime2.guided_sample2(solver, net, clf.model_ema, plan, xt, prior, feed, fix_mask, x_min, x_max, w_cg)
        if out is not None:
            return out
    with torch.no_grad():
        den_emb = None
        if den_comp is not None:
            # FiLM table of ALL steps: one row per step, or one per (step, trajectory) for conditional denoisers
            if runtime._is_chiunet(net):
                cflat = torch.flatten(cond_vec, 1)
                if cflat.shape != (b, den_comp.prog.meta["cond_dim"]):
                    return None
                den_emb = runtime2.chi_film_table(den_comp, net, runtime.device_times(plan, dev), _f32c(cflat, dev), plan=plan)
            elif use_cond:
                if cond_vec.dim() != 2 or cond_vec.shape != (b, den_comp.prog.emb_dim):
                    return None
                den_emb = runtime2.cond_film_table(den_comp, net, runtime.device_times(plan, dev), _f32c(cond_vec, dev))
            else:
                den_emb = runtime2.plan_film_table(den_comp, net, plan, dev)
        t_vec = runtime.device_times(plan, dev)
        temb = _f32c(net.map_noise(t_vec), dev) if gemm_bound is not None else None
        clf_emb0 = _f32c(clf.model_ema.map_noise(t_vec), dev)
        pn = _predicts_noise(plan, solver)
        if edm:       # D + w sigma^2 grad (reference newedm.py:230) as a shift of the raw network output: (w sigma^2 / c_out) grad
            scale = [w_cg * (st.k[2] ** 2) / st.k[1] for st in plan.steps]
        else:
            scale = [(-(w_cg * st.sigma)) if pn else (w_cg * ((st.sigma ** 2) / st.alpha)) for st in plan.steps]
        cg = (ctypes.c_float * len(scale))(*[float(np.float32(v)) for v in scale])
        steps = host_steps(plan)
        noise = feed.many(xt, plan.n_noise)
        xin = _f32c(xt, dev)
        out = torch.empty_like(xin)
        prior_d = _f32c(prior, dev) if fix_mask is not None else None
        if gemm_bound is not None:
            from . import bigbatch
            den_ptr = ctypes.POINTER(CdxUnet2Launch)()               # NULL: the denoiser is the GEMM executor
            gemm_kw = dict(denoiser_gemm=ctypes.cast(ctypes.pointer(gemm_bound.struct), ctypes.c_void_p),
                           denoiser_emb_dim=gemm_bound.struct.emb_dim,
                           denoiser_chunk=bigbatch.CHUNK_OVERRIDE["chiunet"] or bigbatch._chiunet_chunk(b, h, getattr(net, "model_dim", 32), 1))
        else:
            den = runtime2.describe_forward(den_comp, batch=b, emb=den_emb, t_per_wg=den_t,
                                            emb_per_traj=use_cond or runtime._is_chiunet(net))
            den_ptr, gemm_kw = ctypes.pointer(den), {}
        g = CdxGuidedLaunch(denoiser=den_ptr, classifier=ctypes.pointer(bound._struct), steps=steps, cg_scale=cg, **gemm_kw,
                            n_steps=len(plan.steps), batch=b, hd=h * d, predict_noise=int(pn), temb=runtime._ptr(temb),
                            clf_emb0=clf_emb0.data_ptr(), x_in=xin.data_ptr(), prior=runtime._ptr(prior_d),
                            fix_mask=runtime._ptr(fix_mask), noise=runtime._ptr(noise), x_min=runtime._ptr(x_min),
                            x_max=runtime._ptr(x_max), x_out=out.data_ptr())
        lib = _lib()
        need = lib.cdx_guided_workspace_floats(ctypes.byref(g))
        key = (dev, _stream_ptr(dev))          # scratch is ordered by the stream it is used on: one buffer per (device, stream)
        ws = _ws.get(key)
        if ws is None or ws.numel() < need:
            _ws[key] = ws = torch.empty(int(need), dtype=torch.float32, device=dev)
        g.workspace, g.workspace_floats = ws.data_ptr(), ws.numel()
        _check(lib.cdx_guided_run(ctypes.byref(g), _stream_ptr(dev)), "cdx_guided_run")
    return out
