"""Host side of the big-batch executors (``cdx_dit1d_run`` / ``cdx_resmlp_run``, include/cdx.h, csrc/cdx_bigbatch.hip).

DiT1d and IDQLMlp/NewIDQLMlp are served here: their layers are plain GEMMs over M = batch x tokens rows, so the loop of
``sample()`` becomes a stream of tiled-GEMM / LayerNorm / attention / solver-step launches issued by ONE C call.  This
module only marshals pointers: the checkpoint tensors are used in place (PyTorch layouts), the workspace is a cached
device tensor, step records are the same ``plan.Step`` list the fused kernel consumes.
"""
import ctypes
import os
import weakref
from typing import Optional

import torch

from . import runtime
from .runtime import CdxStep, _check, _dense_hd, _f32c, _predicts_noise, _signature, _stream_ptr, load_library

_FP = ctypes.c_void_p
_I = ctypes.c_int32


class CdxSampling(ctypes.Structure):
    _fields_ = [("batch", _I), ("hd", _I), ("emb_dim", _I), ("cond_dim", _I), ("temb", _FP),
                ("steps", ctypes.POINTER(CdxStep)), ("n_steps", _I), ("temb_per_sample", _I), ("predict_noise", _I),
                ("cfg_mode", _I), ("cfg_w", ctypes.c_float), ("cond", _FP), ("x_in", _FP), ("prior", _FP),
                ("fix_mask", _FP), ("noise", _FP), ("x_min", _FP), ("x_max", _FP), ("x_out", _FP), ("workspace", _FP),
                ("workspace_floats", ctypes.c_longlong), ("chunk", _I)]


class CdxDitBlock(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("ada_w", "ada_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w",
                                   "fc2_b")]


class CdxDitCross(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("in_w", "in_b", "out_w", "out_b")]


class CdxDitWeights(ctypes.Structure):
    _fields_ = [("tokens", _I), ("in_dim", _I), ("emb_dim", _I), ("d_model", _I), ("n_heads", _I), ("depth", _I),
                ("x_proj_w", _FP), ("x_proj_b", _FP), ("pos", _FP), ("map0_w", _FP), ("map0_b", _FP), ("map2_w", _FP),
                ("map2_b", _FP), ("blocks", ctypes.POINTER(CdxDitBlock)), ("fin_ada_w", _FP), ("fin_ada_b", _FP),
                ("fin_w", _FP), ("fin_b", _FP), ("cross", ctypes.POINTER(CdxDitCross))]


class CdxPearcetfBlock(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "r1", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "r2")]


class CdxPearcetfWeights(ctypes.Structure):
    _fields_ = [("act_dim", _I), ("To", _I), ("emb_dim", _I), ("te", _I), ("n_heads", _I), ("n_blocks", _I)] + \
               [(n, _FP) for n in ("ae0_w", "ae0_b", "ae2_w", "ae2_b", "a2i_w", "a2i_b", "t2i_w", "t2i_b", "c2i_w", "c2i_b", "cpos")] + \
               [("blocks", ctypes.POINTER(CdxPearcetfBlock)), ("fin_w", _FP), ("fin_b", _FP)]


class CdxResMlpBlock(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("ln_g", "ln_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class CdxResMlpWeights(ctypes.Structure):
    _fields_ = [("x_dim", _I), ("emb_dim", _I), ("obs_dim", _I), ("hidden", _I), ("n_blocks", _I), ("head_mish", _I),
                ("in_w", _FP), ("in_b", _FP), ("blocks", ctypes.POINTER(CdxResMlpBlock)), ("out_w", _FP), ("out_b", _FP)]


class CdxChitfLayer(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("ln1_g", "ln1_b", "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ln2_g", "ln2_b", "ca_in_w",
                                   "ca_in_b", "ca_out_w", "ca_out_b", "ln3_g", "ln3_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b")]


class CdxChitfEncLayer(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("ln1_g", "ln1_b", "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ln2_g", "ln2_b", "ff1_w", "ff1_b",
                                   "ff2_w", "ff2_b")]


class CdxChitfWeights(ctypes.Structure):
    _fields_ = [("Ta", _I), ("To", _I), ("act_dim", _I), ("obs_dim", _I), ("d_model", _I), ("n_heads", _I), ("n_layers", _I),
                ("act_emb_w", _FP), ("act_emb_b", _FP), ("pos_emb", _FP), ("obs_emb_w", _FP), ("obs_emb_b", _FP),
                ("cond_pos_emb", _FP), ("enc0_w", _FP), ("enc0_b", _FP), ("enc2_w", _FP), ("enc2_b", _FP),
                ("layers", ctypes.POINTER(CdxChitfLayer)), ("lnf_g", _FP), ("lnf_b", _FP), ("head_w", _FP), ("head_b", _FP),
                ("self_mask", _FP), ("memory_mask", _FP), ("n_enc_layers", _I), ("enc_layers", ctypes.POINTER(CdxChitfEncLayer))]


class CdxChiUNetBlock(ctypes.Structure):
    _fields_ = [("cin_a", _I), ("cin_b", _I), ("cout", _I), ("groups", _I)] + \
               [(n, _FP) for n in ("w1a", "w1b", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "film_w", "film_b", "wra", "wrb", "br")]


class CdxUnetAttn(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("ln_g", "ln_b", "qkv_w", "out_w", "out_b")] + [("heads", _I), ("dim_head", _I)]


class CdxChiUNetWeights(ctypes.Structure):
    _fields_ = [(n, _I) for n in ("act_dim", "Ta", "cond_dim", "emb_dim", "kernel_size", "n_levels", "cond_predict_scale",
                                  "model_dim", "final_groups", "emb_hidden", "emb_out", "film_ld")] + \
               [(n, _FP) for n in ("map0_w", "map0_b", "map2_w", "map2_b", "gce_w", "gce_b")] + \
               [("blocks", ctypes.POINTER(CdxChiUNetBlock))] + \
               [(n, ctypes.POINTER(ctypes.c_void_p)) for n in ("down_w", "down_b", "up_w_even", "up_w_odd", "up_b")] + \
               [(n, _FP) for n in ("fin_w", "fin_b", "fin_g", "fin_be", "out_w", "out_b")] + \
               [("local_obs_dim", _I), ("lc_down_w", _FP), ("lc_down_b", _FP), ("attn", ctypes.POINTER(CdxUnetAttn))]


_declared = False


def _lib():
    global _declared
    lib = load_library()
    if not _declared:
        lib.cdx_dit1d_workspace_floats.argtypes = [ctypes.POINTER(CdxDitWeights), ctypes.POINTER(CdxSampling)]
        lib.cdx_dit1d_workspace_floats.restype = ctypes.c_longlong
        lib.cdx_dit1d_run.argtypes = [ctypes.POINTER(CdxDitWeights), ctypes.POINTER(CdxSampling), ctypes.c_void_p]
        lib.cdx_dit1d_run.restype = ctypes.c_int
        lib.cdx_pearcetf_workspace_floats.argtypes = [ctypes.POINTER(CdxPearcetfWeights), ctypes.POINTER(CdxSampling)]
        lib.cdx_pearcetf_workspace_floats.restype = ctypes.c_longlong
        lib.cdx_pearcetf_run.argtypes = [ctypes.POINTER(CdxPearcetfWeights), ctypes.POINTER(CdxSampling), ctypes.c_void_p]
        lib.cdx_pearcetf_run.restype = ctypes.c_int
        lib.cdx_chitf_workspace_floats.argtypes = [ctypes.POINTER(CdxChitfWeights), ctypes.POINTER(CdxSampling)]
        lib.cdx_chitf_workspace_floats.restype = ctypes.c_longlong
        lib.cdx_chitf_run.argtypes = [ctypes.POINTER(CdxChitfWeights), ctypes.POINTER(CdxSampling), ctypes.c_void_p]
        lib.cdx_chitf_run.restype = ctypes.c_int
        lib.cdx_chiunet_workspace_floats.argtypes = [ctypes.POINTER(CdxChiUNetWeights), ctypes.POINTER(CdxSampling)]
        lib.cdx_chiunet_workspace_floats.restype = ctypes.c_longlong
        lib.cdx_chiunet_run.argtypes = [ctypes.POINTER(CdxChiUNetWeights), ctypes.POINTER(CdxSampling), ctypes.c_void_p]
        lib.cdx_chiunet_run.restype = ctypes.c_int
        lib.cdx_resmlp_workspace_floats.argtypes = [ctypes.POINTER(CdxResMlpWeights), ctypes.POINTER(CdxSampling)]
        lib.cdx_resmlp_workspace_floats.restype = ctypes.c_longlong
        lib.cdx_resmlp_run.argtypes = [ctypes.POINTER(CdxResMlpWeights), ctypes.POINTER(CdxSampling), ctypes.c_void_p]
        lib.cdx_resmlp_run.restype = ctypes.c_int
        _declared = True
    return lib


# ------------------------------------------------------------------------------------------------ #
# backbone recognition + weight marshalling (cached per module, invalidated when a parameter changes)  #
# ------------------------------------------------------------------------------------------------ #
def is_dit1d(module) -> bool:
    from ..nn_diffusion.dit import DiT1d
    return type(module) is DiT1d


def is_dit1ref(module) -> bool:
    from ..nn_diffusion.dit import DiT1Ref
    return type(module) is DiT1Ref


def is_pearcetf(module) -> bool:
    from ..nn_diffusion.pearcetransformer import PearceTransformer
    return type(module) is PearceTransformer


def is_resmlp(module) -> bool:
    from ..nn_diffusion.mlp_backbones import IDQLMlp, NewIDQLMlp
    return type(module) in (IDQLMlp, NewIDQLMlp)


def is_chitf(module) -> bool:
    from ..nn_diffusion.chitransformer import ChiTransformer
    return type(module) is ChiTransformer


class _Bound:
    """ctypes weight struct + the tensors it points into (kept alive here)."""

    def __init__(self, struct, keep, sig):
        self.struct, self.keep, self.sig = struct, keep, sig


_cache = weakref.WeakKeyDictionary()


def _dev_f32(t: torch.Tensor, keep: list, device):
    """Pointer of an fp32 contiguous tensor on `device`; parameters that already are one are used in place."""
    if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
        t = t.detach().to(device=device, dtype=torch.float32).contiguous()
    keep.append(t)
    return t.data_ptr()


def _bind_dit(net, tokens: int, device) -> Optional[_Bound]:
    d = net.d_model
    heads = net.blocks[0].attn.num_heads if len(net.blocks) else 1
    if tokens > 1024 or d > 1024 or d % heads or d // heads > 64:          # CDX_ATTN_MAX_T
        return None
    cross_mods = list(net.cross_attns) if is_dit1ref(net) else []
    for a in [blk.attn for blk in net.blocks] + cross_mods:
        if a.in_proj_weight is None or a.in_proj_bias is None or a.bias_k is not None or a.add_zero_attn or \
                not a.batch_first or a.num_heads != heads:
            return None
    keep = []
    p = lambda t: _dev_f32(t, keep, device)  # noqa: E731
    cross = None
    if cross_mods:                                   # DiT1Ref (reference dit.py:135-180): one cross-attention in front of every block
        if len(cross_mods) != len(net.blocks):
            return None
        cross = (CdxDitCross * len(cross_mods))()
        for i, a in enumerate(cross_mods):
            cross[i] = CdxDitCross(p(a.in_proj_weight), p(a.in_proj_bias), p(a.out_proj.weight), p(a.out_proj.bias))
        keep.append(cross)
    blocks = (CdxDitBlock * max(len(net.blocks), 1))()
    for i, blk in enumerate(net.blocks):
        ada, fc1, fc2 = blk.adaLN_modulation[1], blk.mlp[0], blk.mlp[3]
        blocks[i] = CdxDitBlock(p(ada.weight), p(ada.bias), p(blk.attn.in_proj_weight), p(blk.attn.in_proj_bias),
                                p(blk.attn.out_proj.weight), p(blk.attn.out_proj.bias), p(fc1.weight), p(fc1.bias),
                                p(fc2.weight), p(fc2.bias))
    pos = net.pos_emb(torch.arange(tokens, device=device)).float().contiguous()     # dit.py:122-125
    fin = net.final_layer
    w = CdxDitWeights(tokens=tokens, in_dim=net.in_dim, emb_dim=net.emb_dim, d_model=d, n_heads=heads,
                      depth=len(net.blocks), x_proj_w=p(net.x_proj.weight), x_proj_b=p(net.x_proj.bias), pos=p(pos),
                      map0_w=p(net.map_emb[0].weight), map0_b=p(net.map_emb[0].bias), map2_w=p(net.map_emb[2].weight),
                      map2_b=p(net.map_emb[2].bias), blocks=blocks, fin_ada_w=p(fin.adaLN_modulation[1].weight),
                      fin_ada_b=p(fin.adaLN_modulation[1].bias), fin_w=p(fin.linear.weight), fin_b=p(fin.linear.bias), cross=cross)
    keep.append(blocks)
    return _Bound(w, keep, None)


def fold_pearcetf(net) -> Optional[dict]:
    """PearceTransformer weights with everything linear folded, in float64 then fp32 (include/cdx.h, cdx_pearcetf_weights):
    input_to_qkv1 into the attention's in_proj, out_proj into attn1_to_fcn, the 1/1.414 residual scaling and the eval-mode
    BatchNorm1d affine maps into the neighbouring Linear, the sine position codes into the token biases.  None: a variant the
    executor does not take (train mode -> BatchNorm1d batch statistics, non-default activations)."""
    import torch.nn as nn
    blocks_m = list(net.transformer_blocks)
    if net.training or not blocks_m:
        return None
    te, td, heads = blocks_m[0].trans_emb_dim, blocks_m[0].transformer_dim, blocks_m[0].nheads
    if te > 64 or td != te * heads or 2 + net.To > 64:
        return None
    if not isinstance(net.act_emb[1], nn.LeakyReLU) or net.act_emb[1].negative_slope != 0.01:
        return None
    f64 = lambda t: t.detach().to("cpu", torch.float64)  # noqa: E731
    out = {"te": te, "td": td, "heads": heads, "blocks": []}
    for blk in blocks_m:
        mha, bn_a, bn_b = blk.multihead_attn1, blk.norm1a, blk.norm1b
        if mha.in_proj_weight is None or mha.in_proj_bias is None or mha.bias_k is not None or mha.add_zero_attn or mha.batch_first or \
                not isinstance(blk.attn1_fcn[1], nn.GELU) or getattr(blk.attn1_fcn[1], "approximate", "none") != "none" or \
                not (bn_a.track_running_stats and bn_b.track_running_stats and bn_a.affine and bn_b.affine):
            return None
        wa, ba = f64(blk.input_to_qkv1.weight), f64(blk.input_to_qkv1.bias)
        wi, bi = f64(mha.in_proj_weight), f64(mha.in_proj_bias)
        sl = [slice(j * td, (j + 1) * td) for j in range(3)]
        s1 = f64(bn_a.weight) / torch.sqrt(f64(bn_a.running_var) + bn_a.eps)
        s2 = f64(bn_b.weight) / torch.sqrt(f64(bn_b.running_var) + bn_b.eps)
        wf, bfc = f64(blk.attn1_to_fcn.weight), f64(blk.attn1_to_fcn.bias)
        wo, bo = f64(mha.out_proj.weight), f64(mha.out_proj.bias)
        r1, r2 = s1 / 1.414, s2 / 1.414
        fc1, fc2 = blk.attn1_fcn[0], blk.attn1_fcn[2]
        out["blocks"].append({
            "qkv_w": torch.cat([wi[q] @ wa[q] for q in sl], 0), "qkv_b": torch.cat([wi[q] @ ba[q] + bi[q] for q in sl], 0),
            "o_w": r1[:, None] * (wf @ wo), "o_b": r1 * (wf @ bo + bfc) + f64(bn_a.bias) - f64(bn_a.running_mean) * s1, "r1": r1,
            "fc1_w": f64(fc1.weight), "fc1_b": f64(fc1.bias),
            "fc2_w": r2[:, None] * f64(fc2.weight), "fc2_b": r2 * f64(fc2.bias) + f64(bn_b.bias) - f64(bn_b.running_mean) * s2, "r2": r2})
    with torch.no_grad():
        dev0 = net.final.weight.device
        pos = lambda v: f64(net.pos_embed(torch.as_tensor(v, dtype=torch.float32, device=dev0).reshape(-1, 1)))  # noqa: E731
        pos12, cpos = pos([1.0, 2.0]), pos([float(3 + j) for j in range(net.To)])
    out.update(a2i_b=f64(net.act_to_input.bias) + pos12[0], t2i_b=f64(net.t_to_input.bias) + pos12[1], cpos=cpos)
    to32 = lambda v: v.to(torch.float32).contiguous() if isinstance(v, torch.Tensor) else v  # noqa: E731
    out["blocks"] = [{k: to32(v) for k, v in b.items()} for b in out["blocks"]]
    return {k: to32(v) for k, v in out.items()}


def _bind_pearcetf(net, device) -> Optional[_Bound]:
    fold = fold_pearcetf(net)
    if fold is None:
        return None
    keep = []
    p = lambda t: _dev_f32(t, keep, device)  # noqa: E731
    arr = (CdxPearcetfBlock * len(fold["blocks"]))()
    for i, b in enumerate(fold["blocks"]):
        arr[i] = CdxPearcetfBlock(*[p(b[k]) for k in ("qkv_w", "qkv_b", "o_w", "o_b", "r1", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "r2")])
    ae0, ae2 = net.act_emb[0], net.act_emb[2]
    w = CdxPearcetfWeights(act_dim=ae0.in_features, To=net.To, emb_dim=net.emb_dim, te=fold["te"], n_heads=fold["heads"],
                           n_blocks=len(fold["blocks"]), ae0_w=p(ae0.weight), ae0_b=p(ae0.bias), ae2_w=p(ae2.weight), ae2_b=p(ae2.bias),
                           a2i_w=p(net.act_to_input.weight), a2i_b=p(fold["a2i_b"]), t2i_w=p(net.t_to_input.weight),
                           t2i_b=p(fold["t2i_b"]), c2i_w=p(net.cond_to_input.weight), c2i_b=p(net.cond_to_input.bias),
                           cpos=p(fold["cpos"]), blocks=arr, fin_w=p(net.final.weight), fin_b=p(net.final.bias))
    keep.append(arr)
    return _Bound(w, keep, None)


def _bind_resmlp(net, device) -> Optional[_Bound]:
    hidden = net.affine_in.out_features
    if hidden > 4096:                                  # cdx_layernorm_f32 keeps a row in registers: C <= 4096
        return None
    keep = []
    p = lambda t: _dev_f32(t, keep, device)  # noqa: E731
    res = list(net.ln_resnet)
    blocks = (CdxResMlpBlock * max(len(res), 1))()
    for i, blk in enumerate(res):
        ln, fc1, fc2 = blk.net[1], blk.net[2], blk.net[4]
        blocks[i] = CdxResMlpBlock(p(ln.weight), p(ln.bias), p(fc1.weight), p(fc1.bias), p(fc2.weight), p(fc2.bias))
    head_mish = isinstance(net.affine_out, torch.nn.Sequential)
    head = net.affine_out[1] if head_mish else net.affine_out
    emb_dim = net.time_mlp[2].out_features
    x_dim = net.affine_in.in_features - emb_dim - net.obs_dim
    w = CdxResMlpWeights(x_dim=x_dim, emb_dim=emb_dim, obs_dim=net.obs_dim, hidden=hidden, n_blocks=len(res),
                         head_mish=int(head_mish), in_w=p(net.affine_in.weight), in_b=p(net.affine_in.bias),
                         blocks=blocks, out_w=p(head.weight), out_b=p(head.bias))
    keep.append(blocks)
    return _Bound(w, keep, None)


def _bind_chitf(net, device) -> Optional[_Bound]:
    import torch.nn as nn
    d = net.act_emb.out_features
    layers = list(net.decoder.layers)
    if net.T > 64 or 1 + net.To > 16 or d > 1024 or net.decoder.norm is not None:
        return None
    enc_layers = []
    if not isinstance(net.encoder, nn.Sequential):    # n_cond_layers > 0: nn.TransformerEncoder (reference chitransformer.py:91-95)
        if not isinstance(net.encoder, nn.TransformerEncoder) or net.encoder.norm is not None:
            return None
        enc_layers = list(net.encoder.layers)
        for lyr in enc_layers:
            if not lyr.norm_first or getattr(lyr.activation, "__name__", "") != "gelu" or lyr.self_attn.in_proj_weight is None or \
                    not lyr.self_attn.batch_first or lyr.self_attn.num_heads != layers[0].self_attn.num_heads:
                return None
    heads = layers[0].self_attn.num_heads if layers else 1
    if d % heads or d // heads > 64:
        return None
    for lyr in layers:
        act = lyr.activation
        if not lyr.norm_first or getattr(act, "__name__", "") != "gelu" or lyr.self_attn.in_proj_weight is None or \
                lyr.multihead_attn.in_proj_weight is None or not lyr.self_attn.batch_first:
            return None
    keep = []
    p = lambda t: _dev_f32(t, keep, device)  # noqa: E731
    arr = (CdxChitfLayer * max(len(layers), 1))()
    for i, l in enumerate(layers):
        sa, ca = l.self_attn, l.multihead_attn
        arr[i] = CdxChitfLayer(p(l.norm1.weight), p(l.norm1.bias), p(sa.in_proj_weight), p(sa.in_proj_bias),
                               p(sa.out_proj.weight), p(sa.out_proj.bias), p(l.norm2.weight), p(l.norm2.bias),
                               p(ca.in_proj_weight), p(ca.in_proj_bias), p(ca.out_proj.weight), p(ca.out_proj.bias),
                               p(l.norm3.weight), p(l.norm3.bias), p(l.linear1.weight), p(l.linear1.bias), p(l.linear2.weight),
                               p(l.linear2.bias))
    enc = (CdxChitfEncLayer * max(len(enc_layers), 1))()
    for i, l in enumerate(enc_layers):
        sa = l.self_attn
        enc[i] = CdxChitfEncLayer(p(l.norm1.weight), p(l.norm1.bias), p(sa.in_proj_weight), p(sa.in_proj_bias), p(sa.out_proj.weight),
                                  p(sa.out_proj.bias), p(l.norm2.weight), p(l.norm2.bias), p(l.linear1.weight), p(l.linear1.bias),
                                  p(l.linear2.weight), p(l.linear2.bias))
    mlp_enc = not enc_layers
    neg = torch.finfo(torch.float32).min               # the kernels clamp scores at -3e38; -inf entries map onto that floor
    w = CdxChitfWeights(Ta=net.T, To=net.To, act_dim=net.act_emb.in_features, obs_dim=net.obs_dim, d_model=d, n_heads=heads,
                        n_layers=len(layers), act_emb_w=p(net.act_emb.weight), act_emb_b=p(net.act_emb.bias),
                        pos_emb=p(net.pos_emb[0]), obs_emb_w=p(net.obs_emb.weight), obs_emb_b=p(net.obs_emb.bias),
                        cond_pos_emb=p(net.cond_pos_emb[0]), enc0_w=p(net.encoder[0].weight) if mlp_enc else None,
                        enc0_b=p(net.encoder[0].bias) if mlp_enc else None, enc2_w=p(net.encoder[2].weight) if mlp_enc else None,
                        enc2_b=p(net.encoder[2].bias) if mlp_enc else None, n_enc_layers=len(enc_layers), enc_layers=enc,
                        layers=arr, lnf_g=p(net.ln_f.weight),
                        lnf_b=p(net.ln_f.bias), head_w=p(net.head.weight), head_b=p(net.head.bias),
                        self_mask=p(net.mask.detach().clamp_min(neg)), memory_mask=p(net.memory_mask.detach().clamp_min(neg)))
    keep += [arr, enc]
    return _Bound(w, keep, None)


def _bind_chiunet(net, Ta: int, device) -> Optional[_Bound]:
    """ChiUNet1d (global conditioning) for the implicit-GEMM executor: conv weights are re-packed (c_out, k, c_in) once."""
    import torch.nn as nn
    from . import blocks as B
    local = not net.obs_as_global_cond                  # local conditioning: one observation row per position (chiunet.py:78-82)
    if (local and net.local_cond_encoder is None) or (not local and net.global_cond_encoder is None):
        return None
    n_levels = len(net.downs)
    if local and n_levels < 2:
        return None
    if Ta & (Ta - 1) or (Ta >> (n_levels - 1)) < 1 or n_levels > 8:
        return None
    keep = []
    p = lambda t: _dev_f32(t, keep, device)  # noqa: E731

    def packed(t):                                      # derived tensors are always fresh fp32 device copies
        t = t.to(device=device, dtype=torch.float32).contiguous()
        keep.append(t)
        return t.data_ptr()

    def bind_block(blk, cin_a, cin_b):
        c1, gn1, c2, gn2 = blk.conv1[0], blk.conv1[1], blk.conv2[0], blk.conv2[1]
        w1 = B.pack_conv(c1.weight)                     # (co, k, ci)
        has_res = isinstance(blk.residual_conv, nn.Conv1d)
        wr = blk.residual_conv.weight.detach()[:, :, 0] if has_res else None
        return CdxChiUNetBlock(
            cin_a=cin_a, cin_b=cin_b, cout=c1.out_channels, groups=gn1.num_groups,
            w1a=packed(w1[:, :, :cin_a]), w1b=packed(w1[:, :, cin_a:]) if cin_b else None, b1=p(c1.bias), g1=p(gn1.weight),
            be1=p(gn1.bias), w2=packed(B.pack_conv(c2.weight)), b2=p(c2.bias), g2=p(gn2.weight), be2=p(gn2.bias),
            film_w=p(blk.cond_encoder[1].weight), film_b=p(blk.cond_encoder[1].bias),
            wra=packed(wr[:, :cin_a]) if has_res else None, wrb=packed(wr[:, cin_a:]) if (has_res and cin_b) else None,
            br=p(blk.residual_conv.bias) if has_res else None)

    blocks = []
    for res1, res2, _ in net.downs:
        blocks += [bind_block(res1, res1.conv1[0].in_channels, 0), bind_block(res2, res2.conv1[0].in_channels, 0)]
    blocks += [bind_block(m, m.conv1[0].in_channels, 0) for m in net.mids]
    for res1, res2, _ in net.ups:
        half = res1.conv1[0].in_channels // 2
        blocks += [bind_block(res1, half, half), bind_block(res2, res2.conv1[0].in_channels, 0)]
    if local:
        enc1, enc2, enc_down = net.local_cond_encoder
        # the two places the local features join have a residual conv in every real net (act_dim != model_dim; concat input)
        if blocks[0].wra is None or blocks[len(blocks) - 2].wra is None:
            return None
        blocks += [bind_block(enc1, enc1.conv1[0].in_channels, 0), bind_block(enc2, enc2.conv1[0].in_channels, 0)]
    for b in blocks:                                    # identity skips need matching widths; the FAST GEMM path wants 16 | c_in
        if b.wra is None and (b.cin_b or b.cin_a != b.cout):
            return None
    arr = (CdxChiUNetBlock * len(blocks))(*blocks)

    def ptr_array(vals):
        a = (ctypes.c_void_p * max(len(vals), 1))(*vals)
        keep.append(a)
        return a
    downs = [lvl[2].conv for lvl in net.downs if not isinstance(lvl[2], nn.Identity)]
    ups = [lvl[2].conv for lvl in net.ups if not isinstance(lvl[2], nn.Identity)]
    if len(downs) != n_levels - 1 or len(ups) != n_levels - 1:
        return None
    up_packed = [B.pack_conv_transpose_k4s2p1(u.weight) for u in ups]
    fin = net.final_conv
    E = net.emb_dim
    w = CdxChiUNetWeights()
    w.act_dim, w.Ta, w.emb_dim = net.downs[0][0].conv1[0].in_channels, Ta, E
    w.cond_dim = 0 if local else net.global_cond_encoder.in_features
    w.kernel_size, w.n_levels, w.cond_predict_scale = fin[0].kernel_size[0], n_levels, int(net.downs[0][0].cond_predict_scale)
    w.model_dim, w.final_groups = net.model_dim, fin[1].num_groups
    w.emb_hidden, w.emb_out, w.film_ld = net.map_emb[0].out_features, E, (E if local else 2 * E)
    w.map0_w, w.map0_b, w.map2_w, w.map2_b = p(net.map_emb[0].weight), p(net.map_emb[0].bias), p(net.map_emb[2].weight), p(net.map_emb[2].bias)
    if local:
        w.gce_w, w.gce_b = None, None
        w.local_obs_dim = net.local_cond_encoder[0].conv1[0].in_channels
        w.lc_down_w, w.lc_down_b = packed(B.pack_conv(net.local_cond_encoder[2].conv.weight)), p(net.local_cond_encoder[2].conv.bias)
    else:
        w.gce_w, w.gce_b = p(net.global_cond_encoder.weight), p(net.global_cond_encoder.bias)
    w.blocks = arr
    w.down_w, w.down_b = ptr_array([packed(B.pack_conv(d.weight)) for d in downs]), ptr_array([p(d.bias) for d in downs])
    w.up_w_even, w.up_w_odd = ptr_array([packed(e) for e, _ in up_packed]), ptr_array([packed(o) for _, o in up_packed])
    w.up_b = ptr_array([p(u.bias) for u in ups])
    w.fin_w, w.fin_b, w.fin_g, w.fin_be = packed(B.pack_conv(fin[0].weight)), p(fin[0].bias), p(fin[1].weight), p(fin[1].bias)
    w.out_w, w.out_b = packed(fin[3].weight.detach()[:, :, 0]), p(fin[3].bias)
    keep.append(arr)
    return _Bound(w, keep, None)


def _bind_janner_gemm(net, H: int, device) -> Optional[_Bound]:
    """Unconditional JannerUNet1d for the same implicit-GEMM executor (bias-only FiLM from Linear(Mish(emb)), no obs half)."""
    import torch.nn as nn
    from . import blocks as B
    from ..utils import GroupNorm1d
    if H & (H - 1):
        return None
    n_levels = len(net.downs)
    if (H >> (n_levels - 1)) < 1 or n_levels > 8:
        return None
    keep = []
    p = lambda t: _dev_f32(t, keep, device)  # noqa: E731

    def packed(t):
        t = t.to(device=device, dtype=torch.float32).contiguous()
        keep.append(t)
        return t.data_ptr()

    def bind_block(blk, cin_a, cin_b):
        c1, gn1, c2, gn2 = blk.conv1[0], blk.conv1[1], blk.conv2[0], blk.conv2[1]
        if not isinstance(gn1, GroupNorm1d) or not isinstance(gn2, GroupNorm1d):
            return None
        w1 = B.pack_conv(c1.weight)
        has_res = isinstance(blk.residual_conv, nn.Conv1d)
        wr = blk.residual_conv.weight.detach()[:, :, 0] if has_res else None
        return CdxChiUNetBlock(
            cin_a=cin_a, cin_b=cin_b, cout=c1.out_channels, groups=gn1.num_groups,
            w1a=packed(w1[:, :, :cin_a]), w1b=packed(w1[:, :, cin_a:]) if cin_b else None, b1=p(c1.bias), g1=p(gn1.weight),
            be1=p(gn1.bias), w2=packed(B.pack_conv(c2.weight)), b2=p(c2.bias), g2=p(gn2.weight), be2=p(gn2.bias),
            film_w=p(blk.emb_mlp[1].weight), film_b=p(blk.emb_mlp[1].bias),
            wra=packed(wr[:, :cin_a]) if has_res else None, wrb=packed(wr[:, cin_a:]) if (has_res and cin_b) else None,
            br=p(blk.residual_conv.bias) if has_res else None)

    blocks = []
    for res1, res2, _, _ in net.downs:
        blocks += [bind_block(res1, res1.conv1[0].in_channels, 0), bind_block(res2, res2.conv1[0].in_channels, 0)]
    blocks += [bind_block(m, m.conv1[0].in_channels, 0) for m in (net.mid_block1, net.mid_block2)]
    for res1, res2, _, _ in net.ups:
        half = res1.conv1[0].in_channels // 2
        blocks += [bind_block(res1, half, half), bind_block(res2, res2.conv1[0].in_channels, 0)]
    if any(b is None for b in blocks):
        return None
    for b in blocks:
        if b.wra is None and (b.cin_b or b.cin_a != b.cout):
            return None
    arr = (CdxChiUNetBlock * len(blocks))(*blocks)

    def ptr_array(vals):
        a = (ctypes.c_void_p * max(len(vals), 1))(*vals)
        keep.append(a)
        return a
    downs = [lvl[3].conv for lvl in net.downs if not isinstance(lvl[3], nn.Identity)]
    ups = [lvl[3].conv for lvl in net.ups if not isinstance(lvl[3], nn.Identity)]
    if len(downs) != n_levels - 1 or len(ups) != n_levels - 1:
        return None
    up_packed = [B.pack_conv_transpose_k4s2p1(u.weight) for u in ups]
    fin = net.final_conv
    if not isinstance(fin[1], GroupNorm1d):
        return None
    w = CdxChiUNetWeights()
    w.act_dim, w.Ta, w.cond_dim, w.emb_dim = net.in_dim, H, 0, net.map_emb[0].in_features
    w.kernel_size, w.n_levels, w.cond_predict_scale = net.kernel_size, n_levels, 0
    if fin[0].kernel_size[0] != net.kernel_size:
        return None                                   # the executor uses one kernel size for the blocks and the final conv
    w.model_dim, w.final_groups = net.model_dim, fin[1].num_groups
    w.emb_hidden, w.emb_out, w.film_ld = net.map_emb[0].out_features, net.map_emb[2].out_features, net.map_emb[2].out_features
    w.map0_w, w.map0_b, w.map2_w, w.map2_b = p(net.map_emb[0].weight), p(net.map_emb[0].bias), p(net.map_emb[2].weight), p(net.map_emb[2].bias)
    w.gce_w, w.gce_b = None, None
    w.blocks = arr
    w.down_w, w.down_b = ptr_array([packed(B.pack_conv(d.weight)) for d in downs]), ptr_array([p(d.bias) for d in downs])
    w.up_w_even, w.up_w_odd = ptr_array([packed(e) for e, _ in up_packed]), ptr_array([packed(o) for _, o in up_packed])
    w.up_b = ptr_array([p(u.bias) for u in ups])
    w.fin_w, w.fin_b, w.fin_g, w.fin_be = packed(B.pack_conv(fin[0].weight)), p(fin[0].bias), p(fin[1].weight), p(fin[1].bias)
    w.out_w, w.out_b = packed(fin[3].weight.detach()[:, :, 0]), p(fin[3].bias)
    if getattr(net, "attention", False):
        # LinearAttention (reference jannerunet.py:72-95) after every level's second block and between the middle blocks: channel
        # LayerNorm -> to_qkv GEMM -> cdx_linattn_f32 -> to_out GEMM + the normalised input
        from ..nn_diffusion.jannerunet import LinearAttention
        sites = [lvl[2] for lvl in net.downs] + [net.mid_attn] + [lvl[2] for lvl in net.ups]
        if len(sites) != 2 * n_levels or any(type(a) is not LinearAttention or a.to_qkv.bias is not None for a in sites):
            return None
        att = (CdxUnetAttn * len(sites))()
        for i, a in enumerate(sites):
            inner = a.to_out.in_channels
            if inner % a.heads or inner // a.heads > 64 or abs(a.scale - (inner // a.heads) ** -0.5) > 1e-12 or abs(a.norm.eps - 1e-5) > 1e-12:
                return None
            att[i] = CdxUnetAttn(p(a.norm.g.reshape(-1)), p(a.norm.b.reshape(-1)), p(a.to_qkv.weight.reshape(a.to_qkv.out_channels, -1)),
                                 p(a.to_out.weight.reshape(a.to_out.out_channels, -1)), p(a.to_out.bias), a.heads, inner // a.heads)
        w.attn = att
        keep.append(att)
    keep.append(arr)
    return _Bound(w, keep, None)


def _bound(net, key, make) -> Optional[_Bound]:
    per_mod = _cache.setdefault(net, {})
    sig = _signature(net)
    hit = per_mod.get(key)
    if hit is not None and hit.sig == sig:
        return hit
    b = make()
    if b is not None:
        b.sig = sig
        per_mod[key] = b
    return b


_workspaces = {}


def _workspace(device, floats: int) -> torch.Tensor:
    """Scratch of the big-batch executors, one buffer per (device, stream): work is ordered by the stream it is enqueued on, so
    two solvers sampling on different torch streams must not share (or regrow) one buffer."""
    key = (device, _stream_ptr(device))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < floats:
        _workspaces[key] = ws = torch.empty(int(floats), dtype=torch.float32, device=device)
    return ws


def host_steps(plan):
    """plan.Step list -> host-resident cdx_step array (the C loop reads it while enqueuing); memoised on the plan."""
    from .plan import cached
    return cached(plan, ("host_steps",), lambda: _pack_host_steps(plan))


def _pack_host_steps(plan):
    arr = (CdxStep * max(len(plan.steps), 1))()
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
    return arr


# Chunk sizes (measured on MI355X, tools/bench_configs.py with CDX_DIT_CHUNK / CDX_MLP_CHUNK): the GEMMs are compute bound, so
# tall chunks win -- more 128 x 128 tiles per launch means fuller last waves of workgroups and fewer launches -- until the
# widest activation (rows x 4 width fp32) outgrows the 256 MiB Infinity Cache by much: DiT1d d=320 peaks at 64 k rows (335 MB),
# IDQLMlp hidden 1024 at 16 k rows (268 MB); twice that is 5-8 % slower, a quarter of it 15 % slower.
def _dit_chunk(batch: int, tokens: int, d_model: int, two: int) -> int:
    rows = max((320 << 20) // (4 * 4 * d_model), 2048)
    return max(min(batch, rows // (tokens * two)), 1)


def _mlp_chunk(batch: int, hidden: int, two: int) -> int:
    return max(min(batch, max((256 << 20) // (4 * 4 * hidden), 2048) // two), 1)


CHUNK_OVERRIDE = {"dit": int(os.environ.get("CDX_DIT_CHUNK", 0)) or None,      # tuning hooks (tools/bench_configs.py, tests)
                  "mlp": int(os.environ.get("CDX_MLP_CHUNK", 0)) or None,
                  "chitf": int(os.environ.get("CDX_CHITF_CHUNK", 0)) or None,
                  "chiunet": int(os.environ.get("CDX_CHIUNET_CHUNK", 0)) or None}
# ChiUNet1d has two native executors: the one-workgroup-per-trajectory program kernel (weights re-streamed by every CU) and the
# implicit-GEMM executor (weights shared by all rows).  The GEMM one wins once batch x length fills the 128 x 128 tiles.
UNET_GEMM_MIN_BATCH = int(os.environ.get("CDX_UNET_GEMM_MIN_BATCH", 96))   # measured: 1.45x at B=128, 1.18x at 256, 2.1x at 1024
# ... and, at ANY batch, once the net is large: the program kernel's step time is the weight stream of one CU (~13 us per MB: 3.7 ms
# per step for the 275 MB config-3 net, whatever the batch), the executor's is ~0.55 ms of dependent launches plus one pass over
# the weights for the whole batch.  Measured (round 3, profiles/r03_chiunet_small_batch.txt), 50-step DDPM at B = 8 / 32 / 64:
# model_dim 256 (68.9 M parameters) 190 / 181 / 178 ms on the program kernel vs 50 / 52 / 60 ms here; model_dim 64 (4.3 M) 16.5 vs
# 33 ms; model_dim 32 (1.1 M) 10 vs 28 ms -- the crossover sits near 10 M parameters.
UNET_GEMM_MIN_PARAMS = int(float(os.environ.get("CDX_UNET_GEMM_MIN_PARAMS", 10e6)))


# JannerUNet1d's channels are narrow (32..256): its GEMM tiles are mostly padding, so the program kernel keeps small and medium
# batches and the GEMM executor only takes over where weight re-streaming dominates (measured crossover, tools/bench_configs.py).
JANNER_GEMM_MIN_BATCH = int(os.environ.get("CDX_JANNER_GEMM_MIN_BATCH", 2048))   # config-2 net: 0.39x at 256, 0.89x at 1024, 1.16x at 3200


def _n_params(module) -> int:
    n = module.__dict__.get("_cdx_n_params")
    if n is None:
        n = module.__dict__["_cdx_n_params"] = sum(p.numel() for p in module.parameters())
    return n


def is_chiunet_gemm(module, batch: int, horizon: Optional[int] = None, edm: bool = False, forward: bool = False) -> bool:
    """Should this U-Net request go to the implicit-GEMM executor?  Yes from the measured crossover batch up, and -- when the
    horizon is given -- for configurations the one-workgroup program kernel cannot hold at all (wide / long nets whose
    activations exceed the LDS plan): those would otherwise drop to the PyTorch executor."""
    from ..nn_diffusion.chiunet import ChiUNet1d
    from ..nn_diffusion.jannerunet import JannerUNet1d
    if type(module) is JannerUNet1d:
        if horizon is not None and not edm and not forward:
            from . import runtime2
            # sampling loops: the program kernel keeps its lead at every batch size (two or three trajectories per workgroup), also for
            # nets that fit only as a compact one-trajectory program (antmaze Diffuser, H = 128 plans; measured at the antmaze size: 10.9 k
            # vs 6.0 k trajectories/s at B = 256, 13.3 k vs 11.5 k at B = 3200).  Stand-alone forwards (`forward`: per-sample timesteps)
            # and EDM plans keep the crossover rule below
            if runtime2.supported(module, horizon) is None:
                return False
        big = batch >= JANNER_GEMM_MIN_BATCH
    elif type(module) is ChiUNet1d and not module.obs_as_global_cond:
        return True                                     # local conditioning: the executor is its only native path
    elif type(module) is ChiUNet1d and module.obs_as_global_cond:
        big = batch >= UNET_GEMM_MIN_BATCH or _n_params(module) >= UNET_GEMM_MIN_PARAMS
    else:
        return False
    if big or horizon is None:
        return big
    from . import runtime2
    if runtime.supported_backbone(module, horizon, edm) is not None:
        return True                                    # no program holds it: the executor at any batch
    return forward and runtime2.compact_only(module, horizon)       # (compact programs serve sampling loops only)


def _bind_unet_gemm(net, tokens: int, dev):
    from ..nn_diffusion.jannerunet import JannerUNet1d
    return _bind_janner_gemm(net, tokens, dev) if type(net) is JannerUNet1d else _bind_chiunet(net, tokens, dev)


def _chiunet_chunk(batch: int, Ta: int, model_dim: int, two: int) -> int:
    rows = max((256 << 20) // (4 * model_dim), 4096)       # widest live activation ~ rows x model_dim
    return max(min(batch, rows // (Ta * two)), 1)


def janner_forward(net, x, noise, condition=None) -> Optional[torch.Tensor]:
    """JannerUNet1d.forward with per-sample timesteps through the implicit-GEMM executor (what a guided / custom loop calls once per
    step when the net is too large for the one-workgroup program kernel, and ``attention=True`` nets at every size).  A condition
    embedding enters the time embedding BEFORE map_emb (reference jannerunet.py:160-164: emb = map_noise(t) + condition), so a
    conditional forward is the same launch on the per-sample rows map_noise(t) + condition (round 5: conditional attention nets)."""
    if x.dim() != 3 or (condition is not None and (condition.dim() != 2 or condition.shape[0] != x.shape[0])):
        return None
    dev = x.device
    b, H, d = x.shape
    bound = _bound(net, ("chiunet", H), lambda: _bind_janner_gemm(net, H, dev))
    if bound is None or d != bound.struct.act_dim:
        return None
    w = bound.struct
    with torch.no_grad():
        temb = net.map_noise(noise)
        if condition is not None:
            if tuple(condition.shape) != (b, temb.shape[-1]):
                return None
            temb = temb.expand(b, -1) + condition
        temb = _f32c(temb.expand(b, -1), dev)
        xin = _f32c(x, dev)
        out = torch.empty_like(xin)
        _run("chiunet", bound, batch=b, hd=H * d, emb_dim=w.emb_dim, cond_dim=0, temb=temb, steps=None, n_steps=0,
             temb_per_sample=1, predict_noise=0, cfg_mode=0, cfg_w=0.0, cond=None, x_in=xin, prior=None, fix_mask=None,
             noise=None, x_min=None, x_max=None, x_out=out,
             chunk=CHUNK_OVERRIDE["chiunet"] or _chiunet_chunk(b, H, net.model_dim if hasattr(net, "model_dim") else 32, 1))
    return out


def _unet_cond_dim(w) -> int:
    """Width of a request's flattened condition: To * obs_dim (global conditioning) or Ta * obs_dim (local: a row per position)."""
    return w.Ta * w.local_obs_dim if w.local_obs_dim > 0 else w.cond_dim


def chiunet_forward(net, x, noise, condition) -> Optional[torch.Tensor]:
    from ..nn_diffusion.jannerunet import JannerUNet1d
    if type(net) is JannerUNet1d:
        return janner_forward(net, x, noise, condition)
    if x.dim() != 3 or condition is None:
        return None
    dev = x.device
    b, Ta, _ = x.shape
    bound = _bound(net, ("chiunet", Ta), lambda: _bind_chiunet(net, Ta, dev))
    if bound is None or x.shape[2] != bound.struct.act_dim:
        return None
    w = bound.struct
    with torch.no_grad():
        cond = _f32c(torch.flatten(condition, 1), dev)
        cond_dim = _unet_cond_dim(w)
        if cond.shape[1] != cond_dim:
            return None
        temb = _f32c(net.map_noise(noise), dev)
        xin = _f32c(x, dev)
        out = torch.empty_like(xin)
        _run("chiunet", bound, batch=b, hd=Ta * w.act_dim, emb_dim=w.emb_dim, cond_dim=cond_dim, temb=temb, steps=None,
             n_steps=0, temb_per_sample=1, predict_noise=0, cfg_mode=1, cfg_w=1.0, cond=cond, x_in=xin, prior=None,
             fix_mask=None, noise=None, x_min=None, x_max=None, x_out=out,
             chunk=CHUNK_OVERRIDE["chiunet"] or _chiunet_chunk(b, Ta, net.model_dim, 1))
    return out


def _run(kind, bound, *, batch, hd, emb_dim, cond_dim, temb, steps, n_steps, temb_per_sample, predict_noise, cfg_mode,
         cfg_w, cond, x_in, prior, fix_mask, noise, x_min, x_max, x_out, chunk):
    lib = _lib()
    pp = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    s = CdxSampling(batch=batch, hd=hd, emb_dim=emb_dim, cond_dim=cond_dim, temb=temb.data_ptr(), steps=steps,
                    n_steps=n_steps, temb_per_sample=int(temb_per_sample), predict_noise=int(predict_noise),
                    cfg_mode=cfg_mode, cfg_w=float(cfg_w), cond=pp(cond), x_in=x_in.data_ptr(), prior=pp(prior),
                    fix_mask=pp(fix_mask), noise=pp(noise), x_min=pp(x_min), x_max=pp(x_max), x_out=x_out.data_ptr(),
                    workspace=None, workspace_floats=0, chunk=chunk)
    size_fn, run_fn = {"dit": (lib.cdx_dit1d_workspace_floats, lib.cdx_dit1d_run),
                       "pearcetf": (lib.cdx_pearcetf_workspace_floats, lib.cdx_pearcetf_run),
                       "mlp": (lib.cdx_resmlp_workspace_floats, lib.cdx_resmlp_run),
                       "chitf": (lib.cdx_chitf_workspace_floats, lib.cdx_chitf_run),
                       "chiunet": (lib.cdx_chiunet_workspace_floats, lib.cdx_chiunet_run)}[kind]
    need = size_fn(ctypes.byref(bound.struct), ctypes.byref(s))
    ws = _workspace(x_in.device, need)
    s.workspace, s.workspace_floats = ws.data_ptr(), ws.numel()
    if batch:
        _check(run_fn(ctypes.byref(bound.struct), ctypes.byref(s), _stream_ptr(x_in.device)), f"cdx_{kind}_run")


# ------------------------------------------------------------------------------------------------ #
# backbone.forward                                                                                     #
# ------------------------------------------------------------------------------------------------ #
def pearcetf_forward(net, x, noise, condition) -> Optional[torch.Tensor]:
    if x.dim() != 2 or condition is None or condition.dim() != 3:
        return None
    if net.training:
        # train mode: BatchNorm1d uses batch statistics (reference pearcetransformer.py:38-39) -- the folded eval-mode weights a cached
        # binding holds do not describe that network (the cache key is the weight signature, not the mode: ADVICE r3)
        return None
    dev = x.device
    bound = _bound(net, "pearcetf", lambda: _bind_pearcetf(net, dev))
    if bound is None or x.shape[1] != bound.struct.act_dim or tuple(condition.shape[1:]) != (net.To, net.emb_dim):
        return None
    w = bound.struct
    with torch.no_grad():
        temb = _f32c(net.map_noise(noise), dev)
        cond = _f32c(torch.flatten(condition, 1), dev)
        xin = _f32c(x, dev)
        out = torch.empty_like(xin)
        _run("pearcetf", bound, batch=x.shape[0], hd=w.act_dim, emb_dim=w.emb_dim, cond_dim=w.To * w.emb_dim, temb=temb, steps=None,
             n_steps=0, temb_per_sample=1, predict_noise=0, cfg_mode=1, cfg_w=1.0, cond=cond, x_in=xin, prior=None, fix_mask=None,
             noise=None, x_min=None, x_max=None, x_out=out, chunk=CHUNK_OVERRIDE["dit"] or _dit_chunk(x.shape[0], 2 + w.To, w.te * w.n_heads, 1))
    return out


def dit_forward(net, x, noise, condition) -> Optional[torch.Tensor]:
    if x.dim() != 3 or x.shape[2] != net.in_dim * (2 if is_dit1ref(net) else 1):
        return None
    b, tokens, _ = x.shape
    dev = x.device
    bound = _bound(net, ("dit", tokens), lambda: _bind_dit(net, tokens, dev))
    if bound is None:
        return None
    with torch.no_grad():
        temb = _f32c(net.map_noise(noise), dev)
        cond = None if condition is None else _f32c(condition, dev)
        xin = _f32c(x, dev)
        out = torch.empty_like(xin)
        _run("dit", bound, batch=b, hd=tokens * x.shape[2], emb_dim=net.emb_dim, cond_dim=net.emb_dim, temb=temb,
             steps=None, n_steps=0, temb_per_sample=1, predict_noise=0, cfg_mode=1 if cond is not None else 0, cfg_w=1.0,
             cond=cond, x_in=xin, prior=None, fix_mask=None, noise=None, x_min=None, x_max=None, x_out=out,
             chunk=CHUNK_OVERRIDE["dit"] or _dit_chunk(b, tokens, net.d_model, 1))
    return out


def chitf_forward(net, x, noise, condition) -> Optional[torch.Tensor]:
    if x.dim() != 3 or x.shape[1] != net.T:
        return None
    dev = x.device
    bound = _bound(net, "chitf", lambda: _bind_chitf(net, dev))
    if bound is None or x.shape[2] != bound.struct.act_dim:
        return None
    w = bound.struct
    with torch.no_grad():
        temb = _f32c(net.map_noise(noise), dev)
        cond = None if condition is None else _f32c(torch.flatten(condition, 1), dev)
        if cond is not None and cond.shape[1] != w.To * w.obs_dim:
            return None
        xin = _f32c(x, dev)
        out = torch.empty_like(xin)
        _run("chitf", bound, batch=x.shape[0], hd=w.Ta * w.act_dim, emb_dim=w.d_model, cond_dim=w.To * w.obs_dim, temb=temb,
             steps=None, n_steps=0, temb_per_sample=1, predict_noise=0, cfg_mode=1 if cond is not None else 0, cfg_w=1.0,
             cond=cond, x_in=xin, prior=None, fix_mask=None, noise=None, x_min=None, x_max=None, x_out=out,
             chunk=CHUNK_OVERRIDE["chitf"] or _dit_chunk(x.shape[0], w.Ta, w.d_model, 1))
    return out


def _time_features(net, t_vec, dev) -> torch.Tensor:
    """time_mlp(map_noise(t)) through the library's own GEMM (rows = number of distinct timesteps)."""
    from . import blocks
    e = _f32c(net.map_noise(t_vec), dev)
    l0, l2 = net.time_mlp[0], net.time_mlp[2]
    h = blocks.linear(e, _f32c(l0.weight, dev), _f32c(l0.bias, dev), act="mish")
    return blocks.linear(h, _f32c(l2.weight, dev), _f32c(l2.bias, dev))


def resmlp_forward(net, x, noise, condition) -> Optional[torch.Tensor]:
    if x.dim() != 2:
        return None
    dev = x.device
    bound = _bound(net, "mlp", lambda: _bind_resmlp(net, dev))
    if bound is None or x.shape[1] != bound.struct.x_dim:
        return None
    with torch.no_grad():
        temb = _time_features(net, noise, dev)
        cond = None if (condition is None or net.obs_dim == 0) else _f32c(condition, dev)
        xin = _f32c(x, dev)
        out = torch.empty_like(xin)
        w = bound.struct
        _run("mlp", bound, batch=x.shape[0], hd=w.x_dim, emb_dim=w.emb_dim, cond_dim=w.obs_dim, temb=temb, steps=None,
             n_steps=0, temb_per_sample=1, predict_noise=0, cfg_mode=1 if cond is not None else 0, cfg_w=1.0, cond=cond,
             x_in=xin, prior=None, fix_mask=None, noise=None, x_min=None, x_max=None, x_out=out,
             chunk=CHUNK_OVERRIDE["mlp"] or _mlp_chunk(x.shape[0], w.hidden, 1))
    return out


# ------------------------------------------------------------------------------------------------ #
# sample()                                                                                             #
# ------------------------------------------------------------------------------------------------ #
def sample(solver, net, plan, xt, prior, cond_vec, w_cfg, feed) -> Optional[torch.Tensor]:
    """Whole denoising loop through cdx_dit1d_run / cdx_resmlp_run.  None -> caller uses the PyTorch executor."""
    dev = xt.device
    if is_pearcetf(net):
        if xt.dim() != 2 or cond_vec is None or w_cfg == 0.0 or cond_vec.dim() != 3:
            return None                               # (the reference cannot run this backbone without a condition either)
        kind, (b, d) = "pearcetf", xt.shape
        if net.training:                 # (see pearcetf_forward: sample(use_ema=False) on a module in train mode)
            return None
        bound = _bound(net, "pearcetf", lambda: _bind_pearcetf(net, dev))
        hd, rows_h = d, 1
        if bound is not None and (bound.struct.act_dim != d or tuple(cond_vec.shape[1:]) != (net.To, net.emb_dim)):
            return None
    elif is_dit1d(net) or is_dit1ref(net):
        if xt.dim() != 3 or xt.shape[2] != net.in_dim * (2 if is_dit1ref(net) else 1):
            return None
        kind, (b, tokens, d) = "dit", xt.shape
        bound = _bound(net, ("dit", tokens), lambda: _bind_dit(net, tokens, dev))
        hd, rows_h = tokens * d, tokens
    elif is_chiunet_gemm(net, xt.shape[0], xt.shape[1] if xt.dim() == 3 else None, runtime.plan_is_edm(plan)):
        from ..nn_diffusion.jannerunet import JannerUNet1d
        janner = type(net) is JannerUNet1d
        if xt.dim() != 3 or (janner and cond_vec is not None and w_cfg != 0.0) or \
                (not janner and (cond_vec is None or w_cfg == 0.0)):
            return None                               # ChiUNet1d needs a condition (the reference raises); Janner here: none
        if janner:
            cond_vec = None
        kind, (b, tokens, d) = "chiunet", xt.shape
        bound = _bound(net, ("chiunet", tokens), lambda: _bind_unet_gemm(net, tokens, dev))
        hd, rows_h = tokens * d, tokens
        if bound is not None and bound.struct.act_dim != d:
            return None
    elif is_chitf(net):
        if xt.dim() != 3 or xt.shape[1] != net.T:
            return None
        kind, (b, tokens, d) = "chitf", xt.shape
        bound = _bound(net, "chitf", lambda: _bind_chitf(net, dev))
        hd, rows_h = tokens * d, tokens
        if bound is not None and bound.struct.act_dim != d:
            return None
    elif is_resmlp(net):
        if xt.dim() != 2:
            return None
        kind, (b, d) = "mlp", xt.shape
        bound = _bound(net, "mlp", lambda: _bind_resmlp(net, dev))
        hd, rows_h = d, 1
        if bound is not None and bound.struct.x_dim != d:
            return None
    else:
        return None
    if bound is None:
        return None
    if cond_vec is None and w_cfg not in (0.0, 1.0):
        return None                                   # the reference raises here; let the torch executor do it
    try:
        fix_mask = _dense_hd(solver.fix_mask, rows_h, d, dev)
        clip = getattr(plan, "clip_each_step", True)
        x_min = _dense_hd(getattr(solver, "x_min", None), rows_h, d, dev) if clip else None
        x_max = _dense_hd(getattr(solver, "x_max", None), rows_h, d, dev) if clip else None
    except (ValueError, RuntimeError):
        return None
    with torch.no_grad():
        t_vec = runtime.device_times(plan, dev)
        if kind == "dit":
            temb, emb_dim, cond_dim = _f32c(net.map_noise(t_vec), dev), net.emb_dim, net.emb_dim
        elif kind == "pearcetf":
            temb, emb_dim, cond_dim = _f32c(net.map_noise(t_vec), dev), net.emb_dim, net.To * net.emb_dim
        elif kind == "chitf":
            temb, emb_dim = _f32c(net.map_noise(t_vec), dev), bound.struct.d_model
            cond_dim = bound.struct.To * bound.struct.obs_dim
        elif kind == "chiunet":
            temb, emb_dim, cond_dim = _f32c(net.map_noise(t_vec), dev), bound.struct.emb_dim, _unet_cond_dim(bound.struct)
        else:
            temb, emb_dim, cond_dim = _time_features(net, t_vec, dev), bound.struct.emb_dim, bound.struct.obs_dim
        if cond_vec is None or w_cfg == 0.0 or (kind == "mlp" and cond_dim == 0):
            mode, cond = 0, None
        else:
            mode, cond = (1 if w_cfg == 1.0 else 2), _f32c(torch.flatten(cond_vec, 1), dev)
            if cond.shape != (b, cond_dim):
                return None
        two = 2 if mode == 2 else 1
        steps = host_steps(plan)
        noise = feed.many(xt, plan.n_noise)
        xin = _f32c(xt, dev)
        out = torch.empty_like(xin)
        if kind == "mlp":
            chunk = CHUNK_OVERRIDE[kind] or _mlp_chunk(b, bound.struct.hidden, two)
        elif kind == "chiunet":
            chunk = CHUNK_OVERRIDE[kind] or _chiunet_chunk(b, rows_h, bound.struct.model_dim, two)
        elif kind == "pearcetf":
            chunk = CHUNK_OVERRIDE["dit"] or _dit_chunk(b, 2 + bound.struct.To, bound.struct.te * bound.struct.n_heads, two)
        else:
            chunk = CHUNK_OVERRIDE[kind] or _dit_chunk(b, rows_h, bound.struct.d_model, two)
        _run(kind, bound, batch=b, hd=hd, emb_dim=emb_dim, cond_dim=cond_dim, temb=temb, steps=steps,
             n_steps=len(plan.steps), temb_per_sample=0, predict_noise=_predicts_noise(plan, solver),
             cfg_mode=mode, cfg_w=w_cfg, cond=cond, x_in=xin, prior=_f32c(prior, dev) if fix_mask is not None else None,
             fix_mask=fix_mask, noise=noise, x_min=x_min, x_max=x_max, x_out=out, chunk=chunk)
    return out
