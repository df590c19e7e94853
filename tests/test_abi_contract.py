"""CPU checks of the drop-in boundary: libcdx.so loads, exports every symbol include/cdx.h declares, the ctypes
mirrors have the C layout, csrc/cdx_ops2.h matches engine/program2.py, and argument validation fails loudly."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_libcdx()
    from cleandiffuser_amd.engine import runtime
    return runtime.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "cdx.h")).read()
    names = re.findall(r"^(?:int|long long|const char\*)\s+(cdx_\w+)\s*\(", hdr, flags=re.M)
    assert set(names) >= {"cdx_abi_version", "cdx_last_error", "cdx_probe_mfma_layout", "cdx_gemm_f32",
                          "cdx_layernorm_f32", "cdx_attention_f32", "cdx_act_f32", "cdx_dit1d_run", "cdx_resmlp_run",
                          "cdx_dit1d_workspace_floats", "cdx_resmlp_workspace_floats", "cdx_gemm_set_trace",
                          "cdx_chitf_run", "cdx_chitf_workspace_floats", "cdx_cross_attention_f32", "cdx_chiunet_run",
                          "cdx_chiunet_workspace_floats", "cdx_groupnorm_f32", "cdx_groupnorm_bwd_f32", "cdx_hjgrad_run",
                          "cdx_hjgrad_workspace_floats", "cdx_guided_run", "cdx_guided_workspace_floats", "cdx_unet2_run",
                          "cdx_unet2_embtab", "cdx_optim_f32", "cdx_pearcetf_run", "cdx_pearcetf_workspace_floats", "cdx_act_bwd_f32", "cdx_linattn_f32",
                          "cdx_conv_wgrad_f32", "cdx_conv_wgrad_batch_f32", "cdx_colsum_f32", "cdx_device_query", "cdx_layernorm_bwd_f32",
                          "cdx_attention_bwd_f32", "cdx_mha_train_fwd_f32", "cdx_mha_train_bwd_f32", "cdx_relayout_f32"}
    for n in names:
        assert hasattr(lib, n), f"{n} declared in cdx.h but not exported by libcdx.so"
    assert lib.cdx_abi_version() == int(re.search(r"#define CDX_ABI_VERSION (\d+)", hdr).group(1))


def test_ctypes_mirrors_have_c_layout(tmp_path):
    from cleandiffuser_amd.engine import bigbatch, blocks, classifier_grad, guided, optim, runtime, runtime2
    mirrors = {"cdx_optim_args": optim.CdxOptimArgs, "cdx_guided_launch": guided.CdxGuidedLaunch, "cdx_unet2_launch": runtime2.CdxUnet2Launch,
               "cdx_unet2_embtab_args": runtime2.CdxUnet2EmbtabArgs,
               "cdx_hj_block": classifier_grad.CdxHjBlock, "cdx_hj_down": classifier_grad.CdxHjDown,
               "cdx_hjgrad_weights": classifier_grad.CdxHjgradWeights,
               "cdx_step": runtime.CdxStep, "cdx_gemm_args": blocks.CdxGemmArgs,
               "cdx_ln_args": blocks.CdxLnArgs, "cdx_attn_args": blocks.CdxAttnArgs, "cdx_sampling": bigbatch.CdxSampling,
               "cdx_dit1d_block": bigbatch.CdxDitBlock, "cdx_dit1d_weights": bigbatch.CdxDitWeights,
               "cdx_dit1ref_cross": bigbatch.CdxDitCross, "cdx_pearcetf_block": bigbatch.CdxPearcetfBlock,
               "cdx_pearcetf_weights": bigbatch.CdxPearcetfWeights,
               "cdx_resmlp_block": bigbatch.CdxResMlpBlock, "cdx_resmlp_weights": bigbatch.CdxResMlpWeights,
               "cdx_chitf_layer": bigbatch.CdxChitfLayer, "cdx_chitf_weights": bigbatch.CdxChitfWeights,
               "cdx_xattn_args": blocks.CdxXattnArgs, "cdx_gn_args": blocks.CdxGnArgs,
               "cdx_chiunet_block": bigbatch.CdxChiUNetBlock, "cdx_chiunet_weights": bigbatch.CdxChiUNetWeights,
               "cdx_unet_attn": bigbatch.CdxUnetAttn, "cdx_wgrad_args": blocks.CdxWgradArgs, "cdx_wgrad_batch": blocks.CdxWgradBatch,
               "cdx_gather_args": blocks.CdxGatherArgs, "cdx_gather_field": blocks.CdxGatherField,
               "cdx_device_props": runtime2.CdxDeviceProps, "cdx_ln_bwd_args": blocks.CdxLnBwdArgs, "cdx_attn_bwd_args": blocks.CdxAttnBwdArgs,
               "cdx_mha_train_args": blocks.CdxMhaTrainArgs, "cdx_relayout_job": blocks.CdxRelayoutJob}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "cdx.h"', 'int main(void){']
    for cname, mirror in mirrors.items():
        src.append(f'printf("%zu\\n", sizeof({cname}));')
        src += [f'printf("%zu\\n", offsetof({cname}, {f[0]}));' for f in mirror._fields_]
    src += ['return 0;}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = iter(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    for cname, mirror in mirrors.items():
        assert int(next(out)) == ctypes.sizeof(mirror), cname
        for f in mirror._fields_:
            assert getattr(mirror, f[0]).offset == int(next(out)), (cname, f[0])
    assert ctypes.sizeof(runtime.CdxStep) == 48


def test_shared_constants_match_headers():
    """Activation ids (include/cdx.h CDX_ACT_*) and the MFMA modes / GroupNorm epsilon of csrc/cdx_ops2.h == engine/consts.py."""
    from cleandiffuser_amd.engine import consts as P
    hdr = open(os.path.join(ROOT, "include", "cdx.h")).read()
    acts = re.findall(r"#define CDX_(ACT_\w+) (\d+)\b", hdr)
    assert len(acts) == 9
    for name, val in acts:
        assert getattr(P, name) == int(val), name
    ops2 = open(os.path.join(ROOT, "cleandiffuser_amd", "csrc", "cdx_ops2.h")).read()
    for name, val in re.findall(r"#define CDX_(MODE_\w+) (\d+)\b", ops2):
        assert getattr(P, name) == int(val), name
    assert abs(float(re.search(r"#define CDX_GN_EPS ([0-9.e+-]+)f", ops2).group(1)) - P.GN_EPS) < 1e-12
    assert not os.path.exists(os.path.join(ROOT, "cleandiffuser_amd", "csrc", "cdx_unet1d.hip"))     # one program kernel


def test_op2_word_layout_matches_header():
    """csrc/cdx_ops2.h (what cdx_unet2.hip decodes) == engine/program2.py (what the host emits)."""
    from cleandiffuser_amd.engine import program2 as P2
    text = open(os.path.join(ROOT, "cleandiffuser_amd", "csrc", "cdx_ops2.h")).read()
    defs = re.findall(r"#define CDX2_(\w+) (\d+)\b", text)
    assert len(defs) > 40
    alias = {"ITEM_WORDS": "ITEM2_WORDS"}
    for name, val in defs:
        name = alias.get(name, name)
        assert hasattr(P2, name), f"program2.py lacks {name}"
        assert getattr(P2, name) == int(val), (name, val, getattr(P2, name))
    hdr = open(os.path.join(ROOT, "include", "cdx.h")).read()
    m = re.search(r"#define CDX2_OP_WORDS\(n_waves\) \((\d+) \+ (\d+) \* \(n_waves\)\)", hdr)
    assert m and all(int(m.group(1)) + int(m.group(2)) * nw == P2.op_words(nw) for nw in (4, 8))
    assert P2.W2_ITEM0 == P2.HDR_WORDS


def test_unet2_validation_fails_loudly(lib):
    from cleandiffuser_amd.engine import runtime2
    runtime2._lib()
    assert lib.cdx_unet2_run(ctypes.byref(runtime2.CdxUnet2Launch()), None) == -1 and b"null" in lib.cdx_last_error()
    L = runtime2.CdxUnet2Launch(ops=8, wblob=8, x_in=8, x_out=8, emb=8, n_ops=1, batch=4, horizon=4, dim=4, traj_floats=64,
                                traj_per_wg=4)
    assert lib.cdx_unet2_run(ctypes.byref(L), None) == -1 and b"traj_per_wg" in lib.cdx_last_error()
    L.traj_per_wg, L.n_waves = 3, 8
    assert lib.cdx_unet2_run(ctypes.byref(L), None) == -1 and b"compact" in lib.cdx_last_error()     # three per workgroup: compact programs only
    L.compact = 1
    assert lib.cdx_unet2_run(ctypes.byref(L), None) == -1 and b"alias" in lib.cdx_last_error()      # compact: x_out is the state storage
    L.compact, L.n_waves = 0, 0
    L.traj_per_wg, L.traj_floats = 2, 30 * 1024
    assert lib.cdx_unet2_run(ctypes.byref(L), None) == -1 and b"n_waves" in lib.cdx_last_error()
    L.n_waves = 8
    assert lib.cdx_unet2_run(ctypes.byref(L), None) == -2 and b"160 KiB" in lib.cdx_last_error()
    # the launch modes added in round 3 refuse what they cannot run before anything is launched
    M = runtime2.CdxUnet2Launch(ops=8, wblob=8, x_in=8, x_out=8, emb=8, n_ops=1, batch=4, horizon=4, dim=4, traj_floats=64,
                                traj_per_wg=1, n_waves=8)
    M.split_k = 3
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"split / grouped program" in lib.cdx_last_error()
    M.split_k, M.xbuf, M.xerr, M.xchg_floats = 4, 8, 8, 6                                           # granules are float4 pairs
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"split / grouped program" in lib.cdx_last_error()
    M.xchg_floats, M.batch = 1024, 72                                                               # 72 trajectories x 4 members > 256 workgroups
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"resident" in lib.cdx_last_error()
    M.split_group, M.batch = 1, 257                                                                 # grouped: 257 trajectories need 288 workgroups
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"resident" in lib.cdx_last_error()
    M.split_k = 0                                                                                   # a group needs its size
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"split_group without split_k" in lib.cdx_last_error()
    M.split_group, M.split_k = 0, 4
    M.batch, M.emb_per_traj = 4, 1                                                                  # conditional programs are not split
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"unconditional" in lib.cdx_last_error()
    M.split_k, M.emb_per_traj, M.mlp, M.traj_per_wg = 0, 0, 1, 2
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"MLP program" in lib.cdx_last_error()
    M.mlp, M.traj_per_wg, M.logp_out = 0, 1, 8                                                      # log_p needs a program with a classifier head
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"logp_out" in lib.cdx_last_error()
    # ABI 14: the repair gate belongs to ORDINARY launches, the fault hook to split / grouped ones
    M.logp_out, M.split_k, M.xchg_floats, M.run_if = None, 4, 1024, 8
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"run_if" in lib.cdx_last_error()
    M.split_k, M.run_if, M.fault = 0, None, 2
    assert lib.cdx_unet2_run(ctypes.byref(M), None) == -1 and b"fault" in lib.cdx_last_error()
    assert lib.cdx_device_query(0, None, None, None) == -1 and b"cdx_device_query" in lib.cdx_last_error()
    L.batch = 0
    assert lib.cdx_unet2_run(ctypes.byref(L), None) == 0                                            # empty request: nothing to do
    assert lib.cdx_unet2_embtab(ctypes.byref(runtime2.CdxUnet2EmbtabArgs()), None) == -1


def test_bigbatch_validation_fails_loudly(lib):
    from cleandiffuser_amd.engine import bigbatch, blocks
    bigbatch._lib(), blocks._lib()
    assert lib.cdx_gemm_f32(ctypes.byref(blocks.CdxGemmArgs(M=4, N=4, K=4)), None) == -1
    assert b"null" in lib.cdx_last_error()
    assert lib.cdx_gemm_f32(ctypes.byref(blocks.CdxGemmArgs(M=0, N=4, K=4)), None) == 0          # empty batch is not an error
    assert lib.cdx_attention_f32(ctypes.byref(blocks.CdxAttnArgs(B=1, T=1025, n_heads=1, head_dim=8, qkv=8, out=8)), None) == -1
    assert lib.cdx_layernorm_f32(ctypes.byref(blocks.CdxLnArgs(M=1, C=8192, x=8, y=8)), None) == -1
    w, s = bigbatch.CdxDitWeights(), bigbatch.CdxSampling()
    assert lib.cdx_dit1d_run(ctypes.byref(w), ctypes.byref(s), None) == -1
    assert lib.cdx_resmlp_run(ctypes.byref(bigbatch.CdxResMlpWeights()), ctypes.byref(s), None) == -1
    assert lib.cdx_last_error() != b""


def test_missing_library_is_a_hard_error(tmp_path):
    from cleandiffuser_amd.engine import runtime
    with pytest.raises(RuntimeError, match="native library not found"):
        runtime.load_library(str(tmp_path / "nope.so"))


def test_newer_entries_validate_before_touching_the_device(lib):
    """Cross-attention, GroupNorm (+ backward), ChiTransformer / U-Net / guided loops: bad requests are refused with a message and
    no HIP call (this runs without a GPU); workspace queries are pure host arithmetic."""
    from cleandiffuser_amd.engine import bigbatch, blocks, classifier_grad, guided
    bigbatch._lib(), blocks._lib()
    bad = -1
    x = blocks.CdxXattnArgs(B=1, T=4, n_obs=16, n_heads=1, head_dim=8, q=8, kv_shared=8, kv_rows=8, out=8)
    assert lib.cdx_cross_attention_f32(ctypes.byref(x), None) == bad and b"memory tokens" in lib.cdx_last_error()
    assert lib.cdx_cross_attention_f32(ctypes.byref(blocks.CdxXattnArgs(B=0, T=4, n_obs=2, n_heads=1, head_dim=8)), None) == 0
    g = blocks.CdxGnArgs(B=2, L=4, C=30, G=8, x=8, y=8, gamma=8, beta=8)
    assert lib.cdx_groupnorm_f32(ctypes.byref(g), None) == bad and b"multiple of G" in lib.cdx_last_error()
    g = blocks.CdxGnArgs(B=2, L=4, C=32, G=8, x=8, y=8, gamma=8, beta=8, residual=8, act=4)          # SiLU has no backward here
    assert lib.cdx_groupnorm_bwd_f32(ctypes.byref(g), None) == bad
    s = bigbatch.CdxSampling()
    assert lib.cdx_chitf_run(ctypes.byref(bigbatch.CdxChitfWeights()), ctypes.byref(s), None) == bad
    assert lib.cdx_pearcetf_run(ctypes.byref(bigbatch.CdxPearcetfWeights()), ctypes.byref(s), None) == bad and b"PearceTransformer" in lib.cdx_last_error()
    assert lib.cdx_pearcetf_workspace_floats(None, None) == -1
    assert lib.cdx_chiunet_run(ctypes.byref(bigbatch.CdxChiUNetWeights()), ctypes.byref(s), None) == bad
    lib.cdx_guided_run.restype = ctypes.c_int
    assert lib.cdx_guided_run(ctypes.byref(guided.CdxGuidedLaunch()), None) == bad and b"null" in lib.cdx_last_error()
    lib.cdx_act_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    assert lib.cdx_act_f32(None, None, 4, 1, None) == bad
    assert lib.cdx_act_f32(8, 8, 0, 1, None) == 0                                                    # nothing to do is fine
    blocks._lib()
    assert lib.cdx_act_bwd_f32(8, 8, 8, 4, 7, 1.0, None) == bad and b"no derivative" in lib.cdx_last_error()   # the Mish-derivative id has none
    assert lib.cdx_layernorm_bwd_f32(ctypes.byref(blocks.CdxLnBwdArgs(M=4, C=8192, x=8, dy=8, dx=8)), None) == bad
    assert lib.cdx_layernorm_bwd_f32(ctypes.byref(blocks.CdxLnBwdArgs(M=4, C=64, x=8, dy=8, dx=8, gamma=8, scale=8, ldx=64, lddy=64, lddx=64)), None) == bad
    assert b"not both" in lib.cdx_last_error()
    assert lib.cdx_layernorm_bwd_f32(ctypes.byref(blocks.CdxLnBwdArgs(M=0, C=64)), None) == 0
    assert lib.cdx_attention_bwd_f32(ctypes.byref(blocks.CdxAttnBwdArgs(B=1, T=65, n_heads=1, head_dim=8, qkv=8, dout=8, dqkv=8)), None) == bad
    assert lib.cdx_attention_bwd_f32(ctypes.byref(blocks.CdxAttnBwdArgs(B=0, T=8, n_heads=1, head_dim=8)), None) == 0
    mha = dict(B=1, Tq=8, Tk=8, n_heads=2, head_dim=8, q=8, k=8, v=8, out=8, dout=8, dq=8, dk=8, dv=8, ldq=16, ldk=16, ldv=16, ldo=16,
               lddq=16, lddk=16, lddv=16)
    for fn in (lib.cdx_mha_train_fwd_f32, lib.cdx_mha_train_bwd_f32):                              # ABI 15
        assert fn(ctypes.byref(blocks.CdxMhaTrainArgs(**{**mha, "Tk": 65})), None) == bad and b"<= 64" in lib.cdx_last_error()
        assert fn(ctypes.byref(blocks.CdxMhaTrainArgs(**{**mha, "ldk": 8})), None) == bad and b"row stride" in lib.cdx_last_error()
        assert fn(ctypes.byref(blocks.CdxMhaTrainArgs(**{**mha, "v": None})), None) == bad
        assert fn(ctypes.byref(blocks.CdxMhaTrainArgs(**{**mha, "B": 0})), None) == 0
        assert fn(None, None) == bad
    assert lib.cdx_mha_train_bwd_f32(ctypes.byref(blocks.CdxMhaTrainArgs(**{**mha, "dk": None})), None) == bad
    assert lib.cdx_relayout_f32(None, None, 0, None) == 0 and lib.cdx_relayout_f32(None, 8, 3, None) == bad and lib.cdx_relayout_f32(8, 8, -1, None) == bad
    gn = dict(B=2, L=4, C=8, G=2, x=8, y=8, gamma=8, beta=8, residual=8, ldx=8, ldy=8, ldr=8, act=1, eps=1e-5)
    assert lib.cdx_groupnorm_bwd_f32(ctypes.byref(blocks.CdxGnArgs(**gn, dgamma_sum=8)), None) == bad and b"go together" in lib.cdx_last_error()
    assert lib.cdx_groupnorm_bwd_f32(ctypes.byref(blocks.CdxGnArgs(**gn, dgamma_sum=8, dbeta_sum=8, dgamma_part=8, dbeta_part=8)), None) == bad
    assert lib.cdx_act_bwd_f32(8, 8, 8, 4, 8, 0.0, None) == bad and lib.cdx_act_bwd_f32(8, None, 8, 4, 4, 1.0, None) == bad
    assert lib.cdx_act_bwd_f32(8, 8, 8, 0, 4, 1.0, None) == 0
    # workspace sizes: host arithmetic, grows with the chunk, independent of the batch beyond the chunk
    blk = (bigbatch.CdxDitBlock * 2)()
    w = bigbatch.CdxDitWeights(tokens=64, in_dim=29, emb_dim=128, d_model=320, n_heads=10, depth=2, blocks=blk)

    def need(batch, chunk, steps):
        return lib.cdx_dit1d_workspace_floats(ctypes.byref(w), ctypes.byref(bigbatch.CdxSampling(batch=batch, hd=64 * 29, emb_dim=128,
                                                                                                 n_steps=steps, cfg_mode=2, chunk=chunk)))
    assert 0 < need(512, 128, 10) < need(512, 256, 10) < need(512, 512, 10) == need(512, 0, 10) == need(512, 4096, 10)
    assert need(100000, 256, 10) == need(512, 256, 10) and need(512, 256, 20) > need(512, 256, 10)
    assert lib.cdx_dit1d_workspace_floats(None, None) == -1


def test_optimiser_entry_validates_before_touching_the_device(lib):
    """cdx_optim_f32 (multi-tensor AdamW / EMA / gradient norm): refused with a message, no HIP call, on malformed requests."""
    from cleandiffuser_amd.engine import optim
    optim._lib()
    assert lib.cdx_optim_f32(None, None) == -1 and b"null" in lib.cdx_last_error()
    assert lib.cdx_optim_f32(ctypes.byref(optim.CdxOptimArgs()), None) == 0                          # nothing to do
    a = optim.CdxOptimArgs(n_tensors=1, n_chunks=1, chunk_elems=4095, chunks=8, numel=8)
    assert lib.cdx_optim_f32(ctypes.byref(a), None) == -1 and b"multiple of 4" in lib.cdx_last_error()
    a = optim.CdxOptimArgs(n_tensors=1, n_chunks=1, chunk_elems=4096, chunks=8, numel=8, mode=optim.OPT_ADAMW, p=8, g=8, m=8)
    assert lib.cdx_optim_f32(ctypes.byref(a), None) == -1 and b"null pointer" in lib.cdx_last_error()
    a = optim.CdxOptimArgs(n_tensors=1, n_chunks=1, chunk_elems=4096, chunks=8, numel=8, mode=optim.OPT_ADAMW, p=8, g=8, m=8, v=8,
                           beta1=0.9, beta2=0.999, bc2_sqrt=0.0)
    assert lib.cdx_optim_f32(ctypes.byref(a), None) == -1 and b"bc2_sqrt" in lib.cdx_last_error()
    a = optim.CdxOptimArgs(n_tensors=1, n_chunks=1, chunk_elems=4096, chunks=8, numel=8, mode=9)
    assert lib.cdx_optim_f32(ctypes.byref(a), None) == -1 and b"unknown mode" in lib.cdx_last_error()
