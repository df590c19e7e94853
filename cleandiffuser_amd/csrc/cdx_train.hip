// cdx_train.hip -- the kernels the TRAINING step of the U-Net denoisers needs on top of the sampling library (the
// forward / backward of DiffusionModel.update(), reference cleandiffuser/diffusion/diffusionsde.py:94-141, basic.py:66,83-86).
//
// The forward and the backward-DATA pass of a Conv1d / ConvTranspose1d are implicit-GEMM convolutions the library already has
// (cdx_gemm_f32 with conv_taps: a backward-data conv is a conv with flipped, transposed weights; a strided one two parity convs), and
// GroupNorm -> Mish forward / backward-data are cdx_groupnorm_f32 / cdx_groupnorm_bwd_f32.  What training adds is everything that sums
// over the (batch x position) rows:
//
//   cdx_conv_wgrad_f32   dW[a][b][t] += sum_{n, m} P[n, m, a] * Q[n, m * stride + t - pad, b]      (a TN GEMM: both operands are read
//                        with the contraction index -- the row -- as the SLOW dimension, which is exactly how the 32x32x2 MFMA wants its
//                        A and B fragments: lanes 0-31 / 32-63 hold 32 consecutive columns of rows k / k + 1; no transposition anywhere).
//                        nn.Conv1d: P = dY, Q = X -> dW[c_out][c_in][t]; nn.ConvTranspose1d(4, 2, 1): P = X, Q = dY -> dW[c_in][c_out][t].
//   cdx_colsum_f32       out[c] += sum_r x[r][c]          (bias gradients; GroupNorm gain / shift gradients from their per-sample partials)
//   cdx_groupnorm_bwd_f32 (csrc/cdx_gemm.hip) grew two optional outputs: per-(sample, channel) partial sums of dz * x_hat and dz.
//   cdx_gather_windows_f32  the batch a training step consumes, gathered from episode arrays resident in HBM (reference: host-side
//                        collation, dataset/d4rl_mujoco_dataset.py:138-151 + a DataLoader + an H2D copy per step): per field and batch item one
//                        contiguous segment copy; HBM-bound byte movement, every field of the batch in ONE launch.
//
// Sums over rows are split over workgroups and combined with float atomics into a buffer the caller zeroed (torch's own conv backward
// does the same; the order is not fixed, gradients of two identical steps agree to ~1e-7 relative).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/cdx.h"

void cdx_set_err(const char* msg);          // cdx_common.hip

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int WG_BK = 16;          // rows (contraction index) per LDS stage
constexpr int WG_MIN_CHUNKS = 4;          // rows per split-K slice >= 16 x this (host heuristic)
constexpr int WG_T = 64;           // output tile: 64 x 64 per workgroup, 4 wave64 each a 32 x 32 MFMA accumulator
constexpr int WG_LD = WG_T + 4;

// one float4 of a row-major (rows x C) matrix at (row, col .. col + 3): zero outside; scalar path when the row stride or the column
// offset is not 16-byte aligned (c_in = 23 for the first layer of config 2)
__device__ __forceinline__ float4 wg_load4(const float* __restrict__ p, long row, int col, int C, int ld, bool row_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok || col >= C) return v;
    const float* q = p + row * (long)ld + col;
    if (((ld | col) & 3) == 0 && col + 3 < C) return *reinterpret_cast<const float4*>(q);
    v.x = q[0];
    if (col + 1 < C) v.y = q[1];
    if (col + 2 < C) v.z = q[2];
    if (col + 3 < C) v.w = q[3];
    return v;
}

// one workgroup: output tile `tile` (64 x 64 of ca x cb), tap `tap`, row slice `slice` of product g
__device__ __forceinline__ void wgrad_tile(const cdx_wgrad_args& g, int tile, int tap, int slice) {
    __shared__ __attribute__((aligned(16))) float Ps[2][WG_BK][WG_LD];
    __shared__ __attribute__((aligned(16))) float Qs[2][WG_BK][WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tb = (g.cb + WG_T - 1) / WG_T;
    const int a0 = (tile / n_tb) * WG_T, b0 = (tile % n_tb) * WG_T;
    const long R = (long)g.batch * g.l_p;
    const int n_chunks = (int)((R + WG_BK - 1) / WG_BK);
    const int per = (n_chunks + g.k_split - 1) / g.k_split;
    const int c_lo = slice * per, c_hi = min(n_chunks, c_lo + per);
    if (c_lo >= c_hi) return;
    // loader: thread -> (row of the chunk, float4 column)
    const int lrow = tid >> 4, lcol = (tid & 15) * 4;
    float4 rp, rq;
    auto fetch = [&](int chunk) {
        const long r = (long)chunk * WG_BK + lrow;
        const bool ok = r < R;
        const int n = ok ? (int)(r / g.l_p) : 0, m = ok ? (int)(r - (long)n * g.l_p) : 0;
        rp = wg_load4(g.p, r, a0 + lcol, g.ca, g.ldp, ok);
        const int qpos = m * g.stride + tap - g.pad;
        rq = wg_load4(g.q, (long)n * g.l_q + qpos, b0 + lcol, g.cb, g.ldq, ok && qpos >= 0 && qpos < g.l_q);
    };
    // bias gradient riding along (db != NULL: p is d loss / d y): the workgroups of tap 0 / first column tile also sum the rows of
    // their P tiles -- every chunk passes through stage() exactly once
    const bool want_db = g.db != nullptr && tap == 0 && b0 == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto stage = [&](int buf) {
        *reinterpret_cast<float4*>(&Ps[buf][lrow][lcol]) = rp;
        *reinterpret_cast<float4*>(&Qs[buf][lrow][lcol]) = rq;
        if (want_db) { bsum.x += rp.x; bsum.y += rp.y; bsum.z += rp.z; bsum.w += rp.w; }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int lr = lane & 31, lk = lane >> 5;
    fetch(c_lo);
    stage(0);
    if (c_lo + 1 < c_hi) fetch(c_lo + 1);
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
        const int buf = (c - c_lo) & 1;
#pragma unroll
        for (int kk = 0; kk < WG_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ps[buf][kk + lk][wm + lr], Qs[buf][kk + lk][wn + lr], acc, 0, 0, 0);
        if (c + 1 < c_hi) {
            stage(buf ^ 1);                       // (the other stage was last read before the barrier that ended chunk c - 1)
            if (c + 2 < c_hi) fetch(c + 2);
        }
        __syncthreads();
    }
    // D fragment of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int a = a0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk, b = b0 + wn + lr;
        if (a < g.ca && b < g.cb) atomicAdd(g.dw + ((long)a * g.cb + b) * g.taps + tap, acc[r]);
    }
    if (want_db) {                                  // 16 row-threads per float4 column -> one sum per column (stage 0 is free now)
        *reinterpret_cast<float4*>(&Ps[0][lrow][lcol]) = bsum;
        __syncthreads();
        if (tid < WG_T && a0 + tid < g.ca) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < WG_BK; ++r) t += Ps[0][r][tid];
            atomicAdd(g.db + a0 + tid, t);
        }
    }
}

__global__ __launch_bounds__(256) void cdx_conv_wgrad_kernel(const cdx_wgrad_args g) {
    wgrad_tile(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// the products of many layers in one launch: the job table is the kernel argument (scalar loads at a wave-uniform index)
__global__ __launch_bounds__(256) void cdx_conv_wgrad_batch_kernel(const cdx_wgrad_batch B) {
    int j = 0;
    while (j + 1 < B.n_jobs && (int)blockIdx.x >= B.wg_start[j + 1]) ++j;
    const cdx_wgrad_args g = B.job[j];
    const int local = (int)blockIdx.x - B.wg_start[j];
    const int n_tiles = ((g.ca + WG_T - 1) / WG_T) * ((g.cb + WG_T - 1) / WG_T);
    const int rest = local / n_tiles;
    wgrad_tile(g, local - rest * n_tiles, rest % g.taps, rest / g.taps);
}

// out[c] += sum_r x[r][c]: 64 columns x 64 rows per workgroup
constexpr int CS_ROWS = 64;
__global__ __launch_bounds__(256) void cdx_colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long R, int C, int ld) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const long r0 = (long)blockIdx.y * CS_ROWS;
    float s = 0.f;
    if (c < C)
        for (long r = r0 + ty; r < min(R, r0 + CS_ROWS); r += 4) s += x[r * ld + c];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) atomicAdd(out + c, (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]));
}


// ---- batch gather ------------------------------------------------------------------------------------------------------
// grid (segment tiles, batch items in groups of GT_ITEMS, fields): a workgroup copies, for GT_ITEMS consecutive batch items, its 1-KiB
// slice of their segments of one field.  Segments start at arbitrary float offsets (width 17 / 11 / 23 ...), so the copy is dword-wide
// and coalesced along the segment; the source rows of a window are consecutive, i.e. a segment is one contiguous run in HBM.
constexpr int GT_ITEMS = 8;
__global__ __launch_bounds__(256) void cdx_gather_kernel(const cdx_gather_args g) {
    const cdx_gather_field f = g.field[blockIdx.z];
    const int seg = f.width * f.steps;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= seg) return;
    const int b0 = blockIdx.y * GT_ITEMS;
#pragma unroll
    for (int i = 0; i < GT_ITEMS; ++i) {
        const int b = b0 + i;
        if (b >= g.batch) break;
        const long r = g.row0[b];
        f.out[(size_t)b * seg + j] = f.src[(size_t)r * f.width + j];
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the row LayerNorm of cdx_layernorm_f32:  y = xhat * w + b,  w = gamma (affine; idqlmlp.py:14) or 1 + scale[m / rows_per_mod]
// (adaLN modulate; dit.py:10-11) or 1.  One wave per row, x and dy in registers:
//   g = dy * w,   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),   dyxhat = dy * xhat  (optional: what the gain / scale gradient sums)
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void cdx_layernorm_bwd_kernel(const cdx_ln_bwd_args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const float* x = a.x + (size_t)row * a.ldx;
    const float* dy = a.dy + (size_t)row * a.lddy;
    float v[NT], g[NT];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = lane + 64 * t;
        v[t] = c < a.C ? x[c] : 0.f;
        s += v[t];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)a.C;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float d = (lane + 64 * t < a.C) ? v[t] - mean : 0.f;
        v[t] = d;
        s2 += d * d;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = 1.0f / sqrtf(s2 / (float)a.C + a.eps);
    const int b = row / a.rows_per_mod;
    float sg = 0.f, sgx = 0.f;
    float* dyx = a.dyxhat ? a.dyxhat + (size_t)row * a.C : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = lane + 64 * t;
        g[t] = 0.f;
        if (c < a.C) {
            v[t] *= rstd;                                  // xhat
            const float d = dy[c];
            const float w = a.gamma ? a.gamma[c] : (a.scale ? 1.0f + a.scale[(size_t)b * a.ldmod + c] : 1.0f);
            g[t] = d * w;
            sg += g[t];
            sgx += g[t] * v[t];
            if (dyx) dyx[c] = d * v[t];
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
    const float mg = sg / (float)a.C, mgx = sgx / (float)a.C;
    float* dx = a.dx + (size_t)row * a.lddx;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = lane + 64 * t;
        if (c < a.C) dx[c] = rstd * (g[t] - mg - v[t] * mgx);
    }
}

// ------------------------------------------------------------------------------------------------
// The softmax(Q K^T scale + mask) -> dropout -> . V core of nn.MultiheadAttention in TRAINING, forward and backward, for <= 64 query
// and <= 64 key tokens: self-attention on packed qkv rows (dit.py:20,34), the causal self-attention and the memory cross-attention of
// nn.TransformerDecoderLayer (chitransformer.py:108-121,148-154).  One workgroup per (sample, head); Q / K / V / dO come through
// separate pointers and row strides, so packed and split projections are the same kernel.  Nothing but the operands is saved by the
// forward: the backward recomputes P.
//   forward:   P = softmax(S),  Pd = P o keep,  O = Pd V
//   backward:  dV = Pd^T dO,  dPd = dO V^T,  dP = dPd o keep,  dS = P o (dP - rowsum(P o dP)) scale,  dQ = dS K,  dK = dS^T Q
// `keep` holds 0 or 1 / (1 - p) per (sample, head, query, key): the dropout mask is DRAWN by the caller (torch's device generator, so
// that seeding and HIP-graph replay behave as for every other draw of the training step) and applied here.
// Plain fp32 FMA loops over LDS tiles (training-only kernels: 0.66 M MAC per (sample, head) at 64 tokens, head_dim 32).
// ------------------------------------------------------------------------------------------------
#define AB_T 64
#define AB_D 64
struct mha_tiles {
    float *Qs, *Ks, *Vs, *dOs, *Ps, *dPs;
    int ldd, ldt;
};
__device__ __forceinline__ mha_tiles mha_carve(float* lds, int Tq, int Tk, int dh, bool bwd) {
    mha_tiles t;
    t.ldd = dh + 1; t.ldt = Tk + 1;
    t.Qs = lds;
    t.Ks = t.Qs + Tq * t.ldd;
    t.Vs = t.Ks + Tk * t.ldd;
    t.dOs = t.Vs + Tk * t.ldd;
    t.Ps = t.dOs + (bwd ? Tq * t.ldd : 0);
    t.dPs = t.Ps + Tq * t.ldt;
    return t;
}
static size_t mha_lds_bytes(int Tq, int Tk, int dh, bool bwd) {
    return (size_t)((bwd ? 2 : 1) * Tq * (dh + 1) + 2 * Tk * (dh + 1) + (bwd ? 2 : 1) * Tq * (Tk + 1)) * sizeof(float);
}
// P(i, .) <- softmax over the Tk scores of row i (4 lanes per row); returns with the rows normalised
__device__ __forceinline__ void mha_row_softmax(float* Ps, int ldt, int Tq, int Tk, int tid) {
    const int i = tid >> 2, q = tid & 3;
    if (i < Tq) {
        float* row = Ps + i * ldt;
        float mx = -3.0e38f;
        for (int j = q; j < Tk; j += 4) mx = fmaxf(mx, row[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float sum = 0.f;
        for (int j = q; j < Tk; j += 4) { const float ex = __expf(row[j] - mx); row[j] = ex; sum += ex; }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float inv = 1.0f / sum;
        for (int j = q; j < Tk; j += 4) row[j] *= inv;
    }
}

template <bool BWD>
__global__ __launch_bounds__(256) void cdx_mha_train_kernel(const cdx_mha_train_args a) {
    extern __shared__ float ab_lds[];
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / a.n_heads, h = bh - b * a.n_heads;
    const int Tq = a.Tq, Tk = a.Tk, dh = a.head_dim;
    const mha_tiles t = mha_carve(ab_lds, Tq, Tk, dh, BWD);
    const int ldd = t.ldd, ldt = t.ldt;
#define Q(i, d) t.Qs[(i) * ldd + (d)]
#define K(i, d) t.Ks[(i) * ldd + (d)]
#define V(i, d) t.Vs[(i) * ldd + (d)]
#define dO(i, d) t.dOs[(i) * ldd + (d)]
#define P(i, j) t.Ps[(i) * ldt + (j)]
#define dP(i, j) t.dPs[(i) * ldt + (j)]
    const size_t qrow = (size_t)b * Tq, krow = (size_t)b * Tk;
    const int hc = h * dh;
    for (int e = tid; e < Tq * dh; e += 256) {
        const int i = e / dh, d = e - i * dh;
        Q(i, d) = a.q[(qrow + i) * a.ldq + hc + d];
        if (BWD) dO(i, d) = a.dout[(qrow + i) * a.ldo + hc + d];
    }
    for (int e = tid; e < Tk * dh; e += 256) {
        const int i = e / dh, d = e - i * dh;
        K(i, d) = a.k[(krow + i) * a.ldk + hc + d];
        V(i, d) = a.v[(krow + i) * a.ldv + hc + d];
    }
    __syncthreads();
    for (int e = tid; e < Tq * Tk; e += 256) {             // S (and dPd)
        const int i = e / Tk, j = e - i * Tk;
        float sacc = 0.f, pacc = 0.f;
        for (int d = 0; d < dh; ++d) {
            sacc = fmaf(Q(i, d), K(j, d), sacc);
            if (BWD) pacc = fmaf(dO(i, d), V(j, d), pacc);
        }
        P(i, j) = sacc * a.scale + (a.mask ? a.mask[i * Tk + j] : 0.f);
        if (BWD) dP(i, j) = pacc;
    }
    __syncthreads();
    mha_row_softmax(t.Ps, ldt, Tq, Tk, tid);
    const float* keep = a.keep ? a.keep + (size_t)bh * Tq * Tk : nullptr;
    if (!BWD) {
        __syncthreads();
        for (int e = tid; e < Tq * dh; e += 256) {         // O = (P o keep) V
            const int i = e / dh, d = e - i * dh;
            float o = 0.f;
            for (int j = 0; j < Tk; ++j) o = fmaf(keep ? P(i, j) * keep[i * Tk + j] : P(i, j), V(j, d), o);
            a.out[(qrow + i) * a.ldo + hc + d] = o;
        }
        return;
    }
    {                                                      // delta + dS: the same 4 lanes per row that normalised it
        const int i = tid >> 2, q = tid & 3;
        if (i < Tq) {
            float dl = 0.f;
            for (int j = q; j < Tk; j += 4) {
                const float g = keep ? dP(i, j) * keep[i * Tk + j] : dP(i, j);      // dP = dPd o keep
                dP(i, j) = g;
                dl = fmaf(P(i, j), g, dl);
            }
            dl += __shfl_xor(dl, 1, 64);
            dl += __shfl_xor(dl, 2, 64);
            for (int j = q; j < Tk; j += 4) {
                const float p = P(i, j);
                dP(i, j) = p * (dP(i, j) - dl) * a.scale;                              // dS
                if (keep) P(i, j) = p * keep[i * Tk + j];                              // Pd, what dV reads
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < Tq * dh; e += 256) {
        const int i = e / dh, d = e - i * dh;
        float dq = 0.f;
        for (int j = 0; j < Tk; ++j) dq = fmaf(dP(i, j), K(j, d), dq);             // dQ[i] = sum_j dS[i][j] K[j]
        a.dq[(qrow + i) * a.lddq + hc + d] = dq;
    }
    for (int e = tid; e < Tk * dh; e += 256) {
        const int i = e / dh, d = e - i * dh;
        float dk = 0.f, dv = 0.f;
        for (int j = 0; j < Tq; ++j) {
            dk = fmaf(dP(j, i), Q(j, d), dk);              // dK[i] = sum_j dS[j][i] Q[j]
            dv = fmaf(P(j, i), dO(j, d), dv);              // dV[i] = sum_j Pd[j][i] dO[j]
        }
        a.dk[(krow + i) * a.lddk + hc + d] = dk;
        a.dv[(krow + i) * a.lddv + hc + d] = dv;
    }
#undef Q
#undef K
#undef V
#undef dO
#undef P
#undef dP
}

static int mha_bad(const char* who, const char* what) {
    char msg[160];
    snprintf(msg, sizeof(msg), "%s%s", who, what);
    cdx_set_err(msg);
    return CDX_EINVAL;
}
static int mha_train_launch(const cdx_mha_train_args* a, bool bwd, void* hip_stream, const char* who) {
    cdx_set_err("");
    if (!a) { return mha_bad(who, ": null argument block"); }
    if (a->B < 0 || a->Tq <= 0 || a->Tq > AB_T || a->Tk <= 0 || a->Tk > AB_T || a->head_dim <= 0 || a->head_dim > AB_D || a->n_heads <= 0) {
        return mha_bad(who, ": Tq <= 64, Tk <= 64 and head_dim <= 64 required");
    }
    if (a->B == 0) return CDX_OK;
    const int dm = a->n_heads * a->head_dim;
    if (!a->q || !a->k || !a->v || (bwd ? (!a->dout || !a->dq || !a->dk || !a->dv) : !a->out)) {
        return mha_bad(who, ": null pointer");
    }
    if (a->ldq < dm || a->ldk < dm || a->ldv < dm || a->ldo < dm || (bwd && (a->lddq < dm || a->lddk < dm || a->lddv < dm))) {
        return mha_bad(who, ": a row stride is shorter than n_heads * head_dim");
    }
    if ((long long)a->B * a->n_heads > 0x7fffffffLL) { return mha_bad(who, ": batch too large for one launch"); }
    const size_t lds = mha_lds_bytes(a->Tq, a->Tk, a->head_dim, bwd);      // <= 100 KB
    static size_t raised[2] = {0, 0};
    const void* fn = bwd ? reinterpret_cast<const void*>(cdx_mha_train_kernel<true>) : reinterpret_cast<const void*>(cdx_mha_train_kernel<false>);
    if (lds > 48 * 1024 && lds > raised[bwd] && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
        raised[bwd] = lds;
    const dim3 grid((unsigned)(a->B * a->n_heads));
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    if (bwd) hipLaunchKernelGGL(cdx_mha_train_kernel<true>, grid, dim3(256), lds, s, *a);
    else hipLaunchKernelGGL(cdx_mha_train_kernel<false>, grid, dim3(256), lds, s, *a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

// ------------------------------------------------------------------------------------------------
// cdx_relayout_f32: one workgroup per (job, chunk); see cdx.h.  The weights of a denoiser are a few MB in total and L2-resident right
// after the optimiser wrote them: what this saves is launches, not bytes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cdx_relayout_kernel(const cdx_relayout_job* __restrict__ jobs, const int32_t* __restrict__ chunks) {
    const int j = chunks[2 * blockIdx.x], ck = chunks[2 * blockIdx.x + 1];
    const cdx_relayout_job job = jobs[j];
    const long n = (long)job.n0 * job.n1 * job.n2;
    const long lo = (long)ck * CDX_RELAYOUT_CHUNK;
    const long hi = lo + CDX_RELAYOUT_CHUNK < n ? lo + CDX_RELAYOUT_CHUNK : n;
    const int n12 = job.n1 * job.n2;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
        const int i0 = (int)(e / n12), r = (int)(e - (long)i0 * n12);
        const int i1 = r / job.n2, i2 = r - i1 * job.n2;
        job.dst[e] = job.src[(long)i0 * job.s0 + (long)i1 * job.s1 + (long)i2 * job.s2];
    }
}

}  // namespace

extern "C" {

int cdx_relayout_f32(const cdx_relayout_job* jobs, const int32_t* chunks, int32_t n_chunks, void* hip_stream) {
    cdx_set_err("");
    if (n_chunks < 0) { cdx_set_err("cdx_relayout_f32: negative chunk count"); return CDX_EINVAL; }
    if (n_chunks == 0) return CDX_OK;
    if (!jobs || !chunks) { cdx_set_err("cdx_relayout_f32: null table"); return CDX_EINVAL; }
    hipLaunchKernelGGL(cdx_relayout_kernel, dim3((unsigned)n_chunks), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), jobs, chunks);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}


int cdx_layernorm_bwd_f32(const cdx_ln_bwd_args* a, void* hip_stream) {
    cdx_set_err("");
    if (!a) { cdx_set_err("cdx_layernorm_bwd_f32: null argument block"); return CDX_EINVAL; }
    if (a->M < 0 || a->C <= 0 || a->C > 4096) { cdx_set_err("cdx_layernorm_bwd_f32: 0 < C <= 4096 required"); return CDX_EINVAL; }
    if (a->M == 0) return CDX_OK;
    if (!a->x || !a->dy || !a->dx) { cdx_set_err("cdx_layernorm_bwd_f32: null pointer"); return CDX_EINVAL; }
    if (a->gamma && a->scale) { cdx_set_err("cdx_layernorm_bwd_f32: gamma (affine) OR scale (modulate), not both"); return CDX_EINVAL; }
    if (a->ldx < a->C || a->lddy < a->C || a->lddx < a->C || (a->scale && (a->rows_per_mod <= 0 || a->ldmod < a->C))) {
        cdx_set_err("cdx_layernorm_bwd_f32: bad leading dimension / rows_per_mod"); return CDX_EINVAL;
    }
    cdx_ln_bwd_args b = *a;
    if (b.rows_per_mod <= 0) b.rows_per_mod = 1;
    auto kern = a->C <= 1024 ? cdx_layernorm_bwd_kernel<16> : (a->C <= 2048 ? cdx_layernorm_bwd_kernel<32> : cdx_layernorm_bwd_kernel<64>);
    hipLaunchKernelGGL(kern, dim3((a->M + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), b);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_attention_bwd_f32(const cdx_attn_bwd_args* a, void* hip_stream) {
    cdx_set_err("");
    if (!a) { cdx_set_err("cdx_attention_bwd_f32: null argument block"); return CDX_EINVAL; }
    if (a->B < 0 || a->T <= 0 || a->T > AB_T || a->head_dim <= 0 || a->head_dim > AB_D || a->n_heads <= 0) {
        cdx_set_err("cdx_attention_bwd_f32: T <= 64 and head_dim <= 64 required"); return CDX_EINVAL;
    }
    if (a->B == 0) return CDX_OK;
    if (!a->qkv || !a->dout || !a->dqkv) { cdx_set_err("cdx_attention_bwd_f32: null pointer"); return CDX_EINVAL; }
    const int dm = a->n_heads * a->head_dim;               // packed rows [q | k | v]: the general kernel on three column blocks
    cdx_mha_train_args g = {};
    g.q = a->qkv; g.k = a->qkv + dm; g.v = a->qkv + 2 * dm;
    g.dout = a->dout;
    g.dq = a->dqkv; g.dk = a->dqkv + dm; g.dv = a->dqkv + 2 * dm;
    g.B = a->B; g.Tq = g.Tk = a->T; g.n_heads = a->n_heads; g.head_dim = a->head_dim;
    g.ldq = g.ldk = g.ldv = g.lddq = g.lddk = g.lddv = 3 * dm; g.ldo = dm;
    g.scale = a->scale;
    return mha_train_launch(&g, true, hip_stream, "cdx_attention_bwd_f32");
}

int cdx_mha_train_fwd_f32(const cdx_mha_train_args* a, void* hip_stream) { return mha_train_launch(a, false, hip_stream, "cdx_mha_train_fwd_f32"); }
int cdx_mha_train_bwd_f32(const cdx_mha_train_args* a, void* hip_stream) { return mha_train_launch(a, true, hip_stream, "cdx_mha_train_bwd_f32"); }

int cdx_conv_wgrad_batch_f32(const cdx_wgrad_batch* b, void* hip_stream) {
    cdx_set_err("");
    if (!b) { cdx_set_err("cdx_conv_wgrad_batch_f32: null pointer"); return CDX_EINVAL; }
    if (b->n_jobs == 0) return CDX_OK;
    if (b->n_jobs < 0 || b->n_jobs > CDX_WGRAD_BATCH || b->wg_start[0] != 0) { cdx_set_err("cdx_conv_wgrad_batch_f32: 1 .. CDX_WGRAD_BATCH jobs, wg_start[0] == 0"); return CDX_EINVAL; }
    for (int j = 0; j < b->n_jobs; ++j) {
        const cdx_wgrad_args* a = &b->job[j];
        if (!a->p || !a->q || !a->dw) { cdx_set_err("cdx_conv_wgrad_batch_f32: null pointer in a job"); return CDX_EINVAL; }
        if (a->batch <= 0 || a->l_p <= 0 || a->l_q <= 0 || a->ca <= 0 || a->cb <= 0 || a->taps <= 0 || a->taps > 16 || a->stride <= 0 || a->pad < 0 ||
            a->ldp < a->ca || a->ldq < a->cb || a->k_split < 1) {
            cdx_set_err("cdx_conv_wgrad_batch_f32: bad shape in a job (k_split >= 1 is the caller's here)"); return CDX_EINVAL;
        }
        const long wgs = (long)((a->ca + WG_T - 1) / WG_T) * ((a->cb + WG_T - 1) / WG_T) * a->taps * a->k_split;
        if (b->wg_start[j + 1] - b->wg_start[j] != wgs) { cdx_set_err("cdx_conv_wgrad_batch_f32: wg_start does not match the jobs' tile counts"); return CDX_EINVAL; }
    }
    hipLaunchKernelGGL(cdx_conv_wgrad_batch_kernel, dim3(b->wg_start[b->n_jobs]), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), *b);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_conv_wgrad_f32(const cdx_wgrad_args* a, void* hip_stream) {
    cdx_set_err("");
    if (!a || !a->p || !a->q || !a->dw) { cdx_set_err("cdx_conv_wgrad_f32: null pointer"); return CDX_EINVAL; }
    if (a->batch == 0) return CDX_OK;
    if (a->batch < 0 || a->l_p <= 0 || a->l_q <= 0 || a->ca <= 0 || a->cb <= 0 || a->taps <= 0 || a->taps > 16 || a->stride <= 0 || a->pad < 0 ||
        a->ldp < a->ca || a->ldq < a->cb || a->k_split < 0) {
        cdx_set_err("cdx_conv_wgrad_f32: bad shape"); return CDX_EINVAL;
    }
    cdx_wgrad_args g = *a;
    const long R = (long)g.batch * g.l_p;
    const int n_chunks = (int)((R + WG_BK - 1) / WG_BK);
    const int tiles = ((g.ca + WG_T - 1) / WG_T) * ((g.cb + WG_T - 1) / WG_T) * g.taps;
    if (g.k_split == 0) {
        // ~1024 workgroups on the 256 CUs, at least WG_MIN_CHUNKS chunks of 16 rows per slice (every slice ends in one float atomic
        // per output element: thin slices buy parallelism with contention).  CDX_WGRAD_MIN_CHUNKS: tuning hook.
        static const int min_chunks = [] { const char* e = getenv("CDX_WGRAD_MIN_CHUNKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : WG_MIN_CHUNKS; }();
        g.k_split = (1024 + tiles - 1) / tiles;
        if (g.k_split > n_chunks / min_chunks) g.k_split = n_chunks / min_chunks;
        if (g.k_split < 1) g.k_split = 1;
    }
    if (g.k_split > 65535) g.k_split = 65535;
    const dim3 grid(((g.ca + WG_T - 1) / WG_T) * ((g.cb + WG_T - 1) / WG_T), g.taps, g.k_split);
    hipLaunchKernelGGL(cdx_conv_wgrad_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), g);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_colsum_f32(const float* x, float* out, long long rows, int32_t cols, int32_t ld, void* hip_stream) {
    cdx_set_err("");
    if (!x || !out) { cdx_set_err("cdx_colsum_f32: null pointer"); return CDX_EINVAL; }
    if (rows < 0 || cols <= 0 || ld < cols) { cdx_set_err("cdx_colsum_f32: bad shape"); return CDX_EINVAL; }
    if (rows == 0) return CDX_OK;
    const dim3 grid((cols + 63) / 64, (unsigned)((rows + CS_ROWS - 1) / CS_ROWS));
    hipLaunchKernelGGL(cdx_colsum_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), x, out, (long)rows, cols, ld);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_gather_windows_f32(const cdx_gather_args* a, void* hip_stream) {
    cdx_set_err("");
    if (!a) { cdx_set_err("cdx_gather_windows_f32: null argument block"); return CDX_EINVAL; }
    if (a->batch < 0 || a->n_fields <= 0 || a->n_fields > CDX_GATHER_MAX_FIELDS || a->rows < 0) {
        cdx_set_err("cdx_gather_windows_f32: bad shape"); return CDX_EINVAL;
    }
    if (a->batch == 0) return CDX_OK;
    if (!a->row0) { cdx_set_err("cdx_gather_windows_f32: null pointer"); return CDX_EINVAL; }
    int seg_max = 0;
    for (int i = 0; i < a->n_fields; ++i) {
        const cdx_gather_field& f = a->field[i];
        if (!f.src || !f.out) { cdx_set_err("cdx_gather_windows_f32: null pointer"); return CDX_EINVAL; }
        if (f.width <= 0 || f.steps <= 0 || f.steps > a->rows || (long long)f.width * f.steps > (1 << 24)) {
            cdx_set_err("cdx_gather_windows_f32: bad field shape"); return CDX_EINVAL;
        }
        if (f.width * f.steps > seg_max) seg_max = f.width * f.steps;
    }
    const dim3 grid((seg_max + 255) / 256, (a->batch + GT_ITEMS - 1) / GT_ITEMS, a->n_fields);
    if (grid.y > 65535) { cdx_set_err("cdx_gather_windows_f32: batch too large for one launch (<= 524280 items)"); return CDX_EINVAL; }
    hipLaunchKernelGGL(cdx_gather_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), *a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

}  // extern "C"
