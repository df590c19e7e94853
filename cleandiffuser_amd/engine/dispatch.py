"""Dispatch rules between the fused gfx950 executors and the PyTorch executor.

Two native executors: the one-workgroup-per-trajectory program kernel (runtime2.py / program2.py: JannerUNet1d, ChiUNet1d below its
crossover, the HalfJannerUNet1d classifier, PearceMlp / DQLMlp / DVInvMlp / MlpNNDiffusion / SfBCUNet) and the big-batch GEMM executors
(bigbatch.py: DiT1d, DiT1Ref, ChiTransformer, PearceTransformer, IDQLMlp / NewIDQLMlp, the U-Nets at large batch or size).

HIP fast path  <=>  tensor lives on a ROCm device  AND  gradients are off  AND  dtype is fp32  AND
the backbone is one the program compiler understands.  Everything else (CPU tensors, autograd, exotic
variants such as ``attention=True``) runs the stock PyTorch modules.  On a ROCm device a *missing or
unloadable* ``libcdx.so`` is a hard error -- there is no silent eager fallback for supported backbones.
"""
from typing import Optional

import torch


def _on_gpu(t: torch.Tensor) -> bool:
    return t.is_cuda


def _try_backbone_forward(module, x, noise, condition) -> Optional[torch.Tensor]:
    """Serve ``backbone.forward`` from the fused program kernel, or return None for the PyTorch path."""
    if not _on_gpu(x) or torch.is_grad_enabled() and _needs_grad(module, x, condition):
        return None
    if x.dtype != torch.float32:
        return None
    from . import bigbatch, runtime
    if bigbatch.is_dit1d(module) or bigbatch.is_dit1ref(module):
        return bigbatch.dit_forward(module, x, noise, condition)
    if bigbatch.is_pearcetf(module):
        return bigbatch.pearcetf_forward(module, x, noise, condition)
    if bigbatch.is_resmlp(module):
        return bigbatch.resmlp_forward(module, x, noise, condition)
    if bigbatch.is_chitf(module):
        return bigbatch.chitf_forward(module, x, noise, condition)
    if bigbatch.is_chiunet_gemm(module, x.shape[0], x.shape[1] if x.dim() == 3 else None, forward=True):
        y = bigbatch.chiunet_forward(module, x, noise, condition)
        if y is not None:
            return y
    return runtime.backbone_forward(module, x, noise, condition)


def _needs_grad(module, x, condition) -> bool:
    if x.requires_grad or (condition is not None and condition.requires_grad):
        return True
    return any(p.requires_grad for p in module.parameters())


def _try_fused_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, requires_grad, feed):
    """Run the whole denoising loop in one launch, or return None for the PyTorch executor."""
    if not _on_gpu(xt) or requires_grad or xt.dtype != torch.float32:
        return None
    if w_cg != 0.0 and solver.classifier is not None:
        from . import guided             # classifier guidance: fused backbone forward + explicit classifier backward per step
        return guided.guided_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, feed)
    from . import bigbatch, runtime
    net = model["diffusion"]
    if bigbatch.is_chiunet_gemm(net, xt.shape[0], xt.shape[1] if xt.dim() == 3 else None, runtime.plan_is_edm(plan)):
        out = bigbatch.sample(solver, net, plan, xt, prior, cond_vec, w_cfg, feed)
        if out is not None:
            return out
    if bigbatch.is_dit1d(net) or bigbatch.is_resmlp(net) or bigbatch.is_chitf(net) or bigbatch.is_dit1ref(net) or bigbatch.is_pearcetf(net):
        return bigbatch.sample(solver, net, plan, xt, prior, cond_vec, w_cfg, feed)     # EDM / consistency kinds included
    return runtime.fused_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, feed)


def _try_fused_raw(solver, model, plan, z, temperature, prior, feed):
    """The common unconditional request before x_T exists: `z` is the initial N(0, I) draw.  When the second-generation U-Net
    kernel takes the request it forms x_T = (z * temperature) * (1 - fix_mask) + prior * fix_mask itself (reference
    diffusionsde.py:509-510), so a steady-state sample() call launches that kernel and nothing else.
    None -> the caller forms x_T with ATen ops and goes through try_fused_sample as before (no draw consumed here)."""
    if not _on_gpu(z) or z.dtype != torch.float32 or z.dim() != 3:
        return None
    from . import bigbatch, runtime
    net = model["diffusion"]
    if not runtime._is_janner(net) or runtime.plan_is_edm(plan):
        return None
    from . import runtime2
    if runtime2.supported(net, z.shape[1]) is not None:
        return None
    if bigbatch.is_chiunet_gemm(net, z.shape[0], z.shape[1], False):
        return None
    return runtime.fused_sample(solver, model, plan, z, prior, None, 0.0, feed, x_scale=float(temperature))


def _try_fused_edm(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, requires_grad, feed, condition_cg=None):
    """``ContinuousEDM.sample``: big-batch executors for the GEMM-shaped backbones, the program kernel for the rest.  Classifier
    guidance (reference newedm.py:217-234: only with a classifier, w_cg != 0 AND a condition_cg) goes through the per-step guided
    executor; without a condition_cg the reference applies no shift, so the request is an unguided one."""
    if not _on_gpu(xt) or requires_grad or xt.dtype != torch.float32:
        return None
    if w_cg != 0.0 and solver.classifier is not None and condition_cg is not None:
        from . import guided
        return guided.guided_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, feed)
    from . import bigbatch, runtime
    net = model["diffusion"]
    if bigbatch.is_resmlp(net) or bigbatch.is_dit1d(net) or bigbatch.is_chitf(net) or bigbatch.is_dit1ref(net) or bigbatch.is_pearcetf(net):
        return bigbatch.sample(solver, net, plan, xt, prior, cond_vec, w_cfg, feed)
    if bigbatch.is_chiunet_gemm(net, xt.shape[0], xt.shape[1] if xt.dim() == 3 else None, runtime.plan_is_edm(plan)):
        out = bigbatch.sample(solver, net, plan, xt, prior, cond_vec, w_cfg, feed)
        if out is not None:
            return out
    return runtime.fused_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, feed)


def _try_fused_legacy_ddpm(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, requires_grad, feed):
    """Legacy ``DDPM`` class: same fused executor, legacy step kinds."""
    if not _on_gpu(xt) or requires_grad or xt.dtype != torch.float32:
        return None
    if solver.classifier is not None and w_cg != 0.0:
        return None
    from . import bigbatch, runtime
    if bigbatch.is_chiunet_gemm(model["diffusion"], xt.shape[0], xt.shape[1] if xt.dim() == 3 else None, runtime.plan_is_edm(plan)):
        out = bigbatch.sample(solver, model["diffusion"], plan, xt, prior, cond_vec, w_cfg, feed)
        if out is not None:
            return out
    return runtime.fused_sample(solver, model, plan, xt, prior, cond_vec, w_cfg, feed)


def _scoped(fn):
    """One request = one signature scope (runtime.signature_scope): the weight signature of a module is computed once per request."""
    import functools

    @functools.wraps(fn)
    def entry(*a, **k):
        from .runtime import signature_scope
        with signature_scope():
            return fn(*a, **k)
    return entry


try_backbone_forward = _scoped(_try_backbone_forward)
try_fused_sample = _scoped(_try_fused_sample)
try_fused_raw = _scoped(_try_fused_raw)
try_fused_edm = _scoped(_try_fused_edm)
try_fused_legacy_ddpm = _scoped(_try_fused_legacy_ddpm)
