// cdx_unet1d.hip -- fused "program" kernel for 1-D temporal U-Net denoisers on gfx950 (MI355X / CDNA4).
//
// Replaces, for JannerUNet1d-structured backbones, the ~170 ATen launches per denoiser forward plus the
// elementwise solver launches of the reference loop (cleandiffuser/diffusion/diffusionsde.py:526-594 driving
// cleandiffuser/nn_diffusion/jannerunet.py:154-201) with ONE launch for the whole sample() call:
//
//   * one workgroup (8 wave64) per trajectory -- trajectories are independent (GroupNorm is per sample);
//   * the trajectory state x, every activation, the skip stack, the time-embedding vectors and the split-K
//     scratch live in LDS for all denoising steps (<= 160 KiB; plan from engine/program.py); HBM sees the
//     initial state once, the per-step noise/prior/mask reads and the final trajectory;
//   * weights are streamed from L2 / MALL (15.8 MB for the north-star config, same stream for every workgroup
//     and every step) as 1-KiB contiguous MFMA-tile records, one global_load_dwordx4 per wave per 16 K-values;
//   * Conv1d / strided Conv1d / ConvTranspose1d = implicit GEMM out[co][n] = sum_K W[co][K] X[K][n] on
//     v_mfma_f32_16x16x4_f32 (exact fp32, fmaf-chain numerics) with co on the MFMA rows and positions on the
//     columns; B operands are ds_read_b128 straight from the channel-last activation slot (2-row zero halo
//     = conv padding, no predicates for stride-1 convs);
//   * bias + split-K reduce + GroupNorm (wave-per-group shuffle reductions) + Mish + FiLM add + residual add
//     are the epilogue of the conv that produced the tile; concat is two source pointers, never materialised;
//   * classifier-free guidance = the program run twice per step into two prediction slots; the clip, the
//     eps<->x0 conversion, the solver update (coefficients frozen on the host, cdx_step) and the fix-mask blend
//     run on the LDS-resident state.
//
// Executable specification / CPU twin of this file: oracle/lane_sim.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cdx.h"
#include "cdx_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CDX_THREADS (CDX_N_WAVES * 64)
#define CDX_PF 4  // weight records in flight per wave

static thread_local char g_err[256] = "";
static void set_err(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)) with tanh(log(1+e^x)) = n / (n + 2), n = e^x (e^x + 2); softplus threshold 20 as ATen
    const float e = expf(fminf(x, 20.0f));
    const float n = e * (e + 2.0f);
    return x > 20.0f ? x : x * n / (n + 2.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// LDS row (halo included) feeding output position `pos` at kernel tap `tap`, or -1 when it contributes zero.
__device__ __forceinline__ int conv_row(int pos, int tap, int cstride, int cpad, int transposed) {
    if (transposed) {
        const int num = pos + cpad - tap;
        if (num % cstride != 0) return -1;
        return num / cstride + CDX_HALO;
    }
    return pos * cstride + tap - cpad + CDX_HALO;
}

struct ConvGeom {
    int taps, cstride, cpad, transposed, l_out;
    int srcA, strideA, ca, srcB, strideB, cb;
};

// K loop of one conv for the column tiles [0, NT): every wave owns work items (ct, ks).
template <int NT>
__device__ __forceinline__ void conv_kloop(const ConvGeom& g, const float* __restrict__ wrec, int n_ct, int ksplit,
                                           int nchunks, float* __restrict__ lds, int scratch, int sstride,
                                           int lane, int wave) {
    const int j = lane & 15, k4 = lane >> 4;
    const int qa = g.taps * g.ca;
    for (int item = wave; item < n_ct * ksplit; item += CDX_N_WAVES) {
        const int ct = item % n_ct, ks = item / n_ct;
        const int q0 = ks * nchunks / ksplit, q1 = (ks + 1) * nchunks / ksplit;
        f32x4 acc0[NT], acc1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc0[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // decode the first chunk of this K range
        int on_b, tap, cc;
        if (q0 < qa) {
            on_b = 0; tap = q0 / g.ca; cc = q0 % g.ca;
        } else {
            on_b = 1; tap = (q0 - qa) / g.cb; cc = (q0 - qa) % g.cb;
        }
        int src = on_b ? g.srcB : g.srcA, sstr = on_b ? g.strideB : g.strideA, ccn = on_b ? g.cb : g.ca;
        int roff[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int pos = nt * 16 + j;
            const int row = pos < g.l_out ? conv_row(pos, tap, g.cstride, g.cpad, g.transposed) : -1;
            roff[nt] = row >= 0 ? row * sstr + 4 * k4 : -1;
        }
        const float4* wp = reinterpret_cast<const float4*>(wrec) + ((size_t)ct * nchunks + q0) * 64 + lane;
        float4 wr[CDX_PF];
#pragma unroll
        for (int u = 0; u < CDX_PF; ++u)
            if (q0 + u < q1) wr[u] = wp[(size_t)u * 64];
        for (int q = q0; q < q1; q += CDX_PF) {
#pragma unroll
            for (int u = 0; u < CDX_PF; ++u) {
                if (q + u < q1) {
                    const float4 a = wr[u];
                    if (q + u + CDX_PF < q1) wr[u] = wp[(size_t)(q + u + CDX_PF - q0) * 64];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (roff[nt] >= 0) bv = *reinterpret_cast<const float4*>(lds + src + roff[nt] + cc * 16);
                        acc0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv.x, acc0[nt], 0, 0, 0);
                        acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv.y, acc1[nt], 0, 0, 0);
                        acc0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv.z, acc0[nt], 0, 0, 0);
                        acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv.w, acc1[nt], 0, 0, 0);
                    }
                    // advance (source, tap, cc)
                    if (++cc == ccn) {
                        cc = 0;
                        if (++tap == g.taps) {
                            tap = 0; on_b = 1;
                            src = g.srcB; sstr = g.strideB; ccn = g.cb;
                        }
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const int pos = nt * 16 + j;
                            const int row = pos < g.l_out ? conv_row(pos, tap, g.cstride, g.cpad, g.transposed) : -1;
                            roff[nt] = row >= 0 ? row * sstr + 4 * k4 : -1;
                        }
                    }
                }
            }
        }
        // D fragment: lane holds rows 4*k4 + r of column j  ->  scratch[ks][n][ct*16 + 4*k4 + r]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + j;
            if (n < g.l_out) {
                const f32x4 d = acc0[nt] + acc1[nt];
                *reinterpret_cast<float4*>(lds + scratch + (ks * g.l_out + n) * sstride + ct * 16 + 4 * k4) =
                    make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

__device__ void conv_op(const int32_t* __restrict__ op, const float* __restrict__ wblob, float* __restrict__ lds,
                        int scratch, int pred_branch_off, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int c_out = op[CDX_W_COUT], c16 = op[CDX_W_COUT16], l_out = op[CDX_W_LOUT];
    const int flags = op[CDX_W_FLAGS];
    const int dst = op[CDX_W_DST] + ((flags & CDX_F_DST_PRED) ? pred_branch_off : 0);
    const int dstride = op[CDX_W_DST_STRIDE], drows = op[CDX_W_DST_ROWS];
    const int ksplit = op[CDX_W_KSPLIT], nchunks = op[CDX_W_NCHUNKS];
    const int n_ct = c16 >> 4, sstride = c16 + 4;

    ConvGeom g;
    g.taps = op[CDX_W_TAPS]; g.cstride = op[CDX_W_CSTRIDE]; g.cpad = op[CDX_W_CPAD];
    g.transposed = op[CDX_W_TRANSPOSED]; g.l_out = l_out;
    g.srcA = op[CDX_W_SRCA]; g.strideA = op[CDX_W_SRCA_STRIDE]; g.ca = op[CDX_W_CA_CHUNKS];
    g.srcB = op[CDX_W_SRCB]; g.strideB = op[CDX_W_SRCB_STRIDE]; g.cb = op[CDX_W_CB_CHUNKS];

    // 1. clear the destination slot (halo rows + pad columns must read as zero for the consumer)
    if (!(flags & CDX_F_ACCUM)) {
        const int total = drows * dstride;  // multiple of 4, dst 16-byte aligned
        for (int i = tid * 4; i < total; i += CDX_THREADS * 4)
            *reinterpret_cast<float4*>(lds + dst + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // 2. implicit-GEMM K loop -> split-K partials in scratch
    const float* wrec = wblob + op[CDX_W_WOFF];
    const int n_nt = (l_out + 15) >> 4;
    if (n_nt == 1) conv_kloop<1>(g, wrec, n_ct, ksplit, nchunks, lds, scratch, sstride, lane, wave);
    else if (n_nt == 2) conv_kloop<2>(g, wrec, n_ct, ksplit, nchunks, lds, scratch, sstride, lane, wave);
    else conv_kloop<4>(g, wrec, n_ct, ksplit, nchunks, lds, scratch, sstride, lane, wave);
    __syncthreads();

    // 3. epilogue
    const float* bias = wblob + op[CDX_W_BOFF];
    const int res = op[CDX_W_RES], rstride = op[CDX_W_RES_STRIDE];
    if (flags & CDX_F_GN_MISH) {
        const int groups = op[CDX_W_GROUPS];
        const int cg = c_out / groups, cnt = cg * l_out;
        const float* gamma = wblob + op[CDX_W_GAMMA];
        const float* beta = wblob + op[CDX_W_BETA];
        const float inv_cnt = 1.0f / (float)cnt;
        for (int gi = wave; gi < groups; gi += CDX_N_WAVES) {
            float s = 0.f;
            for (int e = lane; e < cnt; e += 64) {
                const int n = e / cg, c = gi * cg + (e - n * cg);
                float v = bias[c];
                for (int ks = 0; ks < ksplit; ++ks) v += lds[scratch + (ks * l_out + n) * sstride + c];
                lds[scratch + n * sstride + c] = v;  // owned by this lane only
                s += v;
            }
            const float mean = wave_sum(s) * inv_cnt;
            float s2 = 0.f;
            for (int e = lane; e < cnt; e += 64) {
                const int n = e / cg, c = gi * cg + (e - n * cg);
                const float d = lds[scratch + n * sstride + c] - mean;
                s2 += d * d;
            }
            const float var = wave_sum(s2) * inv_cnt;
            const float rstd = 1.0f / sqrtf(var + CDX_GN_EPS);
            for (int e = lane; e < cnt; e += 64) {
                const int n = e / cg, c = gi * cg + (e - n * cg);
                float v = (lds[scratch + n * sstride + c] - mean) * rstd * gamma[c] + beta[c];
                v = mish_f(v);
                if (flags & CDX_F_ADD_EMB) v += lds[op[CDX_W_EMB] + c];
                if (flags & CDX_F_ADD_RES) v += lds[res + (n + CDX_HALO) * rstride + c];
                lds[dst + (n + CDX_HALO) * dstride + c] = v;
            }
        }
    } else {
        const int total = c_out * l_out;
        for (int e = tid; e < total; e += CDX_THREADS) {
            const int n = e / c_out, c = e - n * c_out;
            float v = bias[c];
            for (int ks = 0; ks < ksplit; ++ks) v += lds[scratch + (ks * l_out + n) * sstride + c];
            if (flags & CDX_F_ADD_EMB) v += lds[op[CDX_W_EMB] + c];
            if (flags & CDX_F_ADD_RES) v += lds[res + (n + CDX_HALO) * rstride + c];
            const int o = dst + (n + CDX_HALO) * dstride + c;
            if (flags & CDX_F_ACCUM) v += lds[o];
            lds[o] = v;
        }
    }
    __syncthreads();
}

__device__ void run_program(const cdx_unet1d_launch& L, float* __restrict__ lds, int step, int branch, bool use_cond,
                            int b, int tid) {
    for (int oi = 0; oi < L.n_ops; ++oi) {
        const int32_t* op = L.ops + (size_t)oi * CDX_OP_WORDS;
        const int kind = op[CDX_W_KIND];
        if (kind == CDX_OP_CONV) {
            conv_op(op, L.wblob, lds, L.scratch_off, branch * L.pred_branch_floats, tid);
        } else if (kind == CDX_OP_LINEAR) {
            const int n_in = op[CDX_L_NIN], n_out = op[CDX_L_NOUT];
            const float* w = L.wblob + op[CDX_L_WOFF];   // [n_in][n_out]
            const float* bb = L.wblob + op[CDX_L_BOFF];
            const int src = op[CDX_L_SRC], dst = op[CDX_L_DST];
            const bool post = op[CDX_L_FLAGS] & CDX_F_POST_MISH;
            for (int o = tid; o < n_out; o += CDX_THREADS) {
                float acc = bb[o];
                for (int i = 0; i < n_in; ++i) acc = fmaf(w[(size_t)i * n_out + o], lds[src + i], acc);
                lds[dst + o] = post ? mish_f(acc) : acc;
            }
            __syncthreads();
        } else {  // CDX_OP_LOAD_TEMB
            const int n = op[CDX_L_NIN], dst = op[CDX_L_DST];
            for (int i = tid; i < n; i += CDX_THREADS) {
                float v = L.temb[(size_t)(L.temb_per_sample ? b : step) * L.emb_dim + i];
                if (use_cond) v += L.cond[(size_t)b * L.emb_dim + i];
                lds[dst + i] = v;
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(CDX_THREADS) void cdx_unet1d_kernel(const cdx_unet1d_launch L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int H = L.horizon, D = L.dim, HD = H * D;
    const size_t xbase = (size_t)b * HD;

    // ---- state slot: zero (halo + pad channels), then load x_T ----
    {
        const int total = (H + 2 * CDX_HALO) * L.x_stride;
        for (int i = tid * 4; i < total; i += CDX_THREADS * 4)
            *reinterpret_cast<float4*>(lds + L.x_off + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    for (int e = tid; e < HD; e += CDX_THREADS) {
        const int n = e / D, c = e - n * D;
        lds[L.x_off + (n + CDX_HALO) * L.x_stride + c] = L.x_in[xbase + e];
    }
    __syncthreads();

    const int n_iter = L.n_steps > 0 ? L.n_steps : 1;
    for (int step = 0; step < n_iter; ++step) {
        const int n_branch = (L.cfg_mode == 2) ? 2 : 1;
        for (int br = 0; br < n_branch; ++br) {
            const bool use_cond = (L.cond != nullptr) && (L.cfg_mode == 1 || (L.cfg_mode == 2 && br == 0));
            run_program(L, lds, step, br, use_cond, b, tid);
        }
        if (L.n_steps == 0) {  // forward-only: emit the prediction
            for (int e = tid; e < HD; e += CDX_THREADS) {
                const int n = e / D, c = e - n * D;
                L.x_out[xbase + e] = lds[L.pred_off + (n + CDX_HALO) * L.pred_stride + c];
            }
            return;
        }
        // ---- guidance combine, clip, eps/x0 conversion, solver update, fix-mask blend (all on LDS state) ----
        const cdx_step st = L.steps[step];
        const float al = st.alpha, sg = st.sigma;
        const float k0 = st.k[0], k1 = st.k[1], k2 = st.k[2], k3 = st.k[3], k4 = st.k[4];
        for (int e = tid; e < HD; e += CDX_THREADS) {
            const int n = e / D, c = e - n * D;
            const int xo = L.x_off + (n + CDX_HALO) * L.x_stride + c;
            const int po = L.pred_off + (n + CDX_HALO) * L.pred_stride + c;
            const float x = lds[xo];
            float p = lds[po];
            if (L.cfg_mode == 2) p = L.cfg_w * p + (1.0f - L.cfg_w) * lds[po + L.pred_branch_floats];
            if (L.predict_noise) {
                if (L.x_max) p = fmaxf(p, (x - al * L.x_max[e]) / sg);
                if (L.x_min) p = fminf(p, (x - al * L.x_min[e]) / sg);
            } else {
                if (L.x_min) p = fmaxf(p, L.x_min[e]);
                if (L.x_max) p = fminf(p, L.x_max[e]);
            }
            float eps, xth;
            if (L.predict_noise) {
                eps = p; xth = (x - sg * p) / al;
            } else {
                xth = p; eps = (x - al * p) / sg;
            }
            float xn;
            if (st.kind == 0) {
                xn = k0 * (x - k1 * eps) + k2 * eps;
                if (st.noise_idx >= 0) xn += k3 * L.noise[((size_t)st.noise_idx * L.batch + b) * HD + e];
            } else if (st.kind == 1) {
                xn = k0 * ((x - k1 * eps) / k2) + k3 * eps;
            } else {
                float v = st.vsel == 0 ? eps : xth;
                if (st.vsel == 2) v = k3 * xth - k4 * lds[L.prev_off + e];
                xn = k0 * x - k1 * v;
                if (st.noise_idx >= 0) xn += k2 * L.noise[((size_t)st.noise_idx * L.batch + b) * HD + e];
            }
            if (L.fix_mask) {
                const float m = L.fix_mask[e];
                xn = xn * (1.0f - m) + L.prior[xbase + e] * m;
            }
            if (st.push) lds[L.prev_off + e] = xth;
            lds[xo] = xn;
        }
        __syncthreads();
    }
    for (int e = tid; e < HD; e += CDX_THREADS) {
        const int n = e / D, c = e - n * D;
        L.x_out[xbase + e] = lds[L.x_off + (n + CDX_HALO) * L.x_stride + c];
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA layout probe (test hook)
// ------------------------------------------------------------------------------------------------
__global__ void cdx_probe_kernel(float* out) {
    // out[4][64][4]:  0: 16x16x4 A-probe, 1: 16x16x4 B-probe, 2: 4x4x1(16 blocks) A-probe, 3: 4x4x1 B-probe.
    // A-probe: a = digit code of the lane, b = 1  ->  D tells which lanes' A values reach each D element.
    const int l = threadIdx.x;
    const float code16 = (float)((l & 15) + 1) * (float)(1 << (6 * (l >> 4)));  // (i+1) * 64^k, exact in fp32
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(code16, 1.0f, z, 0, 0, 0);
    const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, code16, z, 0, 0, 0);
    const f32x4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);
    const f32x4 d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        out[(0 * 64 + l) * 4 + r] = d0[r];
        out[(1 * 64 + l) * 4 + r] = d1[r];
        out[(2 * 64 + l) * 4 + r] = d2[r];
        out[(3 * 64 + l) * 4 + r] = d3[r];
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int cdx_abi_version(void) { return CDX_ABI_VERSION; }

const char* cdx_last_error(void) { return g_err; }

int cdx_unet1d_run(const cdx_unet1d_launch* L, void* hip_stream) {
    g_err[0] = 0;
    if (!L || !L->ops || !L->wblob || !L->x_in || !L->x_out || !L->temb) { set_err("null pointer in launch"); return CDX_EINVAL; }
    if (L->n_ops <= 0 || L->batch <= 0 || L->horizon <= 0 || L->dim <= 0 || L->emb_dim <= 0) { set_err("non-positive size"); return CDX_EINVAL; }
    if (L->n_steps > 0 && !L->steps) { set_err("steps == NULL with n_steps > 0"); return CDX_EINVAL; }
    if (L->fix_mask && !L->prior) { set_err("fix_mask given without prior"); return CDX_EINVAL; }
    if (L->cfg_mode < 0 || L->cfg_mode > 2) { set_err("cfg_mode must be 0, 1 or 2"); return CDX_EINVAL; }
    if (L->cfg_mode == 2 && !L->cond) { set_err("cfg_mode 2 needs cond"); return CDX_EINVAL; }
    if ((L->x_off | L->pred_off | L->prev_off | L->scratch_off | L->x_stride | L->pred_stride | L->pred_branch_floats) & 3) {
        set_err("LDS offsets/strides must be multiples of 4 floats"); return CDX_EINVAL;
    }
    const size_t lds_bytes = (size_t)L->lds_floats * sizeof(float);
    if (lds_bytes > 160u * 1024u) { set_err("program needs more than 160 KiB of LDS"); return CDX_ELDS; }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cdx_unet1d_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return CDX_EHIP; }
    hipLaunchKernelGGL(cdx_unet1d_kernel, dim3(L->batch), dim3(CDX_THREADS), lds_bytes,
                       reinterpret_cast<hipStream_t>(hip_stream), *L);
    e = hipGetLastError();
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_probe_mfma_layout(float* out_device, void* hip_stream) {
    g_err[0] = 0;
    if (!out_device) { set_err("null output"); return CDX_EINVAL; }
    hipLaunchKernelGGL(cdx_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(hip_stream), out_device);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

}  // extern "C"
