// cdx_gemm.hip -- batched fp32 "Linear" GEMM for the big-batch denoisers (DiT1d, wide MLPs) on gfx950.
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] )        A: (M, K) row-major, W: (N, K) row-major
//
// i.e. exactly ``torch.nn.functional.linear(A, W, bias)`` with the PyTorch weight layout, so the reference
// checkpoints' tensors are used in place (no packing).  These layers are the hot ops of
// cleandiffuser/nn_diffusion/dit.py:31-36,49 and idqlmlp.py:12-18 when the batch is large (M = batch x tokens >> 256):
// there the right shape is a classic tiled GEMM, not the one-workgroup-per-trajectory program kernel.
//
// Kernel: 128 x 128 x 16 block tile of v_mfma_f32_32x32x2_f32 blocks (exact fp32) -- 8 wave64 with 2 x 1 blocks each (round 5, the
// default where loads are unguarded) or 4 wave64 with 2 x 2 each; 64 x 64 x 32 tiles (4 waves, one block each) for small problems.
// K is summed in BLOCKS (16 / 32 k values per block sum, block sums added to a running total) instead of one sequential fma chain
// (round 5: every GEMM at or below ATen's error against float64).  A and W tiles are fetched K-contiguous (float4 per lane), transposed through LDS into
// [k][row] so MFMA operand reads are conflict-free ds_read_b32, next tile prefetched into registers under the MFMAs.
// Epilogue (fused, per element): + bias[n] -> activation -> * gate[m / rows_per_gate][n] -> + residual[m][n]
// -> + table[m % table_rows][n]  (adaLN gates, residual streams and the positional table of DiT never take a pass of
// their own).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/cdx.h"
#include "cdx_ops2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GM_BM 128
#define GM_BN 128
#define GM_BK 16
#define GM_LD (GM_BM + 4)
#define GM_THREADS 256
#ifndef GM_STAGE_AT
#define GM_STAGE_AT (BK - 4)                   // k pair of the first half behind which tile t + 1 is parked in LDS (A/B builds)
#endif
#ifndef CDX_GEMM_W8_BK
#define CDX_GEMM_W8_BK 16                      // K tile of the 8-wave shape: 16 (block sums over two tiles) or 32 (one tile = one block; dynamic LDS)
#endif
#ifndef CDX_GEMM_W8_DEFAULT
#define CDX_GEMM_W8_DEFAULT 1               // the 8-wave shape of the 128 x 128 tile by default (same-box A/B: profiles/r05_gemm_w8_ab.txt)
#endif
#ifndef CDX_GEMM_PERSIST
#define CDX_GEMM_PERSIST 0                    // 1: the 8-wave kernel walks several tiles per workgroup (grid = resident slots), the next tile's first
#endif                                        //    K tile requested before the epilogue of the current one (A/B builds; round 6)
#ifndef CDX_GEMM_KBLOCK
#define CDX_GEMM_KBLOCK 1                     // 0: one sequential fma chain over K per element (rounds 1-4; A/B builds)
#endif

extern void cdx_set_err(const char* msg);

__device__ __forceinline__ float gm_act(float x, int act) {
    switch (act) {
        case CDX_ACT_MISH: {
            const float e = __expf(fminf(x, 20.0f));
            const float n = e * (e + 2.0f);
            return x > 20.0f ? x : x * n * __builtin_amdgcn_rcpf(n + 2.0f);
        }
        case CDX_ACT_GELU_ERF: {                         // exact GELU; erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7, branch-free:
            const float z = fabsf(x) * 0.70710678118654752f;   // libm erff costs ~4x as much in a GEMM epilogue)
            const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
            const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            const float erf_abs = 1.0f - poly * __expf(-z * z);
            return 0.5f * x * (1.0f + copysignf(erf_abs, x));
        }
        case CDX_ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
        case CDX_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
        case CDX_ACT_RELU: return fmaxf(x, 0.f);
        case CDX_ACT_MISH_GRAD: {                        // d/dx [x tanh(softplus x)] = t + x (1 - t^2) sigmoid(x)
            const float e = __expf(fminf(x, 20.0f));
            const float n = e * (e + 2.0f);
            const float t = x > 20.0f ? 1.0f : n / (n + 2.0f);
            const float sg = e / (1.0f + e);
            return t + x * (1.0f - t * t) * sg;
        }
        case CDX_ACT_GELU_TANH: {                        // 0.5 x (1 + tanh u) == x * sigmoid(2u): one v_exp, one v_rcp, no branches
            const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);
            return x * __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
        }
        case CDX_ACT_TANH: {                             // sign(x) (1 - e^{-2|x|}) / (1 + e^{-2|x|}): no overflow, no branches
            const float t = __expf(-2.0f * fabsf(x));
            return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
        }
        default: return x;
    }
}

// load 4 consecutive K values of one row (zero beyond the matrix edge) -- general path only
__device__ __forceinline__ float4 gm_load4_guarded(const float* __restrict__ base, int row, int rows, int k, int K, int ld) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
        const float* p = base + (size_t)row * ld + k;
        if (k < K) v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
        if (k + 3 < K) v.w = p[3];
    }
    return v;
}

// implicit-GEMM conv gather of 4 consecutive k (general path: any conv_cin; one division per element, tiny first layers only)
__device__ __forceinline__ float4 gm_conv_load4(const cdx_gemm_args& g, int row, int k) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < g.M) {
        const int b = row / g.conv_lout, lo = row - b * g.conv_lout;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k + i;
            if (kk < g.K) {
                const int tap = kk / g.conv_cin, c = kk - tap * g.conv_cin;
                const int pos = lo * g.conv_stride + tap - g.conv_pad;
                if (pos >= 0 && pos < g.conv_lin) v[i] = g.A[((size_t)b * g.conv_lin + pos) * g.lda + c];
            }
        }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// exact m / d and m % d for 0 <= m < 2^31 from a float reciprocal (one multiply + two fix-ups instead of a ~40-instruction
// integer division per output element)
__device__ __forceinline__ void gm_divmod(int m, int d, float inv, int& q, int& r) {
    q = (int)((float)m * inv);
    r = m - q * d;
    if (r < 0) { r += d; --q; }
    if (r >= d) { r -= d; ++q; }
}

template <int ACT>
__device__ __forceinline__ float gm_act_t(float x) { return gm_act(x, ACT); }

// Epilogue of one wave's (32 WT) x (32 WT) sub-tile.  D fragment of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
template <int ACT, int WTM, int WTN>
__device__ __forceinline__ void gm_epilogue(const cdx_gemm_args& g, const f32x16 (&acc)[WTM][WTN], int row0, int col0, int lr, int lk) {
    const float inv_gate = g.gate ? 1.0f / (float)g.rows_per_gate : 0.f;
    const float inv_tab = g.table ? 1.0f / (float)g.table_rows : 0.f;
#pragma unroll
    for (int ni = 0; ni < WTN; ++ni) {
        const int n = col0 + ni * 32 + lr;
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < WTM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= g.M) continue;
                float v = gm_act_t<ACT>(acc[mi][ni][r] + bias);
                if (g.gate) {
                    int q, rem;
                    gm_divmod(m, g.rows_per_gate, inv_gate, q, rem);
                    v *= g.gate[(size_t)q * g.ldg + n];
                }
                if (g.residual) v += g.residual[(size_t)m * g.ldr + n];
                if (g.table) {
                    int q, rem;
                    gm_divmod(m, g.table_rows, inv_tab, q, rem);
                    v += g.table[(size_t)rem * g.N + n];
                }
                g.C[(size_t)m * g.ldc + n] = v;
            }
        }
    }
}

// Fast epilogue (N, ldc, ldr, ldg multiples of 4; 16-byte aligned bases): each 32 x 32 MFMA block goes through a wave-private LDS
// patch so that a lane ends up with 4 consecutive columns -> gate / residual / table are read and C is written with 16-byte
// accesses, loads are unconditional (clamped addresses) and issued together, only the store is predicated.
#define GM_EP_LD 36
template <int ACT, int WTM, int WTN>
__device__ __forceinline__ void gm_epilogue_fast(const cdx_gemm_args& g, const f32x16 (&acc)[WTM][WTN], float* __restrict__ patch,
                                                 int row0, int col0, int lane) {
    const int lr = lane & 31, lk = lane >> 5;
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
    const float inv_gate = g.gate ? 1.0f / (float)g.rows_per_gate : 0.f;
    const float inv_tab = g.table ? 1.0f / (float)g.table_rows : 0.f;
#pragma unroll
    for (int ni = 0; ni < WTN; ++ni) {
        const int nb = col0 + ni * 32;
        const float bias = (g.bias && nb + lr < g.N) ? g.bias[nb + lr] : 0.f;
        const int n = nb + pc4;
        const bool n_ok = n < g.N;
        const int nc = n_ok ? n : 0;
#pragma unroll
        for (int mi = 0; mi < WTM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * GM_EP_LD + lr] = gm_act_t<ACT>(acc[mi][ni][r] + bias);
#pragma unroll
            for (int jh = 0; jh < 4; jh += 2) {            // two rows per lane at a time keeps the epilogue under the K loop's VGPRs
                __builtin_amdgcn_sched_barrier(0);         // and the scheduler must not hoist later blocks' loads above this one
                float4 v[2], gt[2], rs[2], tb[2];
                int mrow[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    mrow[j] = row0 + mi * 32 + prow + 8 * (jh + j);
                    const int mc = min(mrow[j], g.M - 1);
                    if (g.gate) {
                        int q, rem;
                        gm_divmod(mc, g.rows_per_gate, inv_gate, q, rem);
                        gt[j] = *reinterpret_cast<const float4*>(g.gate + (size_t)q * g.ldg + nc);
                    }
                    if (g.residual) rs[j] = *reinterpret_cast<const float4*>(g.residual + (size_t)mc * g.ldr + nc);
                    if (g.table) {
                        int q, rem;
                        gm_divmod(mc, g.table_rows, inv_tab, q, rem);
                        tb[j] = *reinterpret_cast<const float4*>(g.table + (size_t)rem * g.N + nc);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    v[j] = *reinterpret_cast<const float4*>(patch + (prow + 8 * (jh + j)) * GM_EP_LD + pc4);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (g.gate) { v[j].x *= gt[j].x; v[j].y *= gt[j].y; v[j].z *= gt[j].z; v[j].w *= gt[j].w; }
                    if (g.residual) { v[j].x += rs[j].x; v[j].y += rs[j].y; v[j].z += rs[j].z; v[j].w += rs[j].w; }
                    if (g.table) { v[j].x += tb[j].x; v[j].y += tb[j].y; v[j].z += tb[j].z; v[j].w += tb[j].w; }
                    if (n_ok && mrow[j] < g.M) *reinterpret_cast<float4*>(g.C + (size_t)mrow[j] * g.ldc + n) = v[j];
                }
            }
        }
    }
}

template <int ACT, int WTM, int WTN>
__device__ __forceinline__ void gm_epilogue_any(const cdx_gemm_args& g, const f32x16 (&acc)[WTM][WTN], float* patch, int row0, int col0,
                                                int lane, bool fast) {
    if (fast) gm_epilogue_fast<ACT, WTM, WTN>(g, acc, patch, row0, col0, lane);
    else gm_epilogue<ACT, WTM, WTN>(g, acc, row0, col0, lane & 31, lane >> 5);
}

// total += block sum, as 8 packed adds (v_pk_add_f32) per 32 x 32 block
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gm_flush(f32x16& total, const f32x16& blk) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 t = f32x2{total[r], total[r + 1]} + f32x2{blk[r], blk[r + 1]};
        total[r] = t[0];
        total[r + 1] = t[1];
    }
}

// Optional timeline trace (tools/gemm_trace.py): [blockIdx][4] x u64 = {s_memtime at start, first tile landed, K loop done,
// epilogue done}, lane 0 only; off unless cdx_gemm_set_trace() installed a buffer.
__device__ unsigned long long* gm_trace = nullptr;
__device__ __forceinline__ void gm_stamp(int slot) {
    if (gm_trace != nullptr && threadIdx.x == 0) gm_trace[(size_t)blockIdx.x * 4 + slot] = __builtin_amdgcn_s_memtime();
}

// FAST: K a multiple of the K tile, 16-byte aligned rows -> unguarded global_load_dwordx4 (rows beyond the edge are clamped: they
// only feed outputs that are never stored).  !FAST: fully guarded scalar loads (K = 29 input projections and the like).
// WT = MFMA tiles per wave per dimension: 2 -> 128 x 128 x 16 workgroup tile (throughput shape), 1 -> 64 x 64 x 32 (few rows:
// four times the workgroups, a quarter of the MFMA time per barrier -- the launches of classifier guidance and small batches).
// NW = wave64 per workgroup: 4 (each wave a WT x WT grid of 32 x 32 blocks), or -- round 5, WT = 2 only -- 8: each wave 2 x 1 blocks of the
// same 128 x 128 tile.  The 8-wave shape exists for the K-blocked accumulation: a wave's block accumulators (32 registers) AND its totals
// (32) fit 128 registers, so two workgroups = four waves per SIMD stay resident, the block sums run over TWO K tiles (32 k values)
// before they are flushed -- half the flushes per MFMA -- and the flush bubbles of one wave are covered by three others.
template <bool FAST, int WT, bool CONV, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : (WT == 2 ? 3 : 4))) void cdx_gemm_kernel(const cdx_gemm_args g, const int fast_ep,
                                                                                             const int k_split, const int xcd_order) {
    static_assert(NW == 4 || (NW == 8 && WT == 2), "8 waves: the 128 x 128 tile only");
    constexpr int THREADS = 64 * NW;
    constexpr int WTM = WT, WTN = NW == 8 ? 1 : WT;      // 32 x 32 blocks per wave (rows x columns)
    constexpr int BMN = 64 * WT;                  // rows of A == rows of W per tile
    constexpr int BK = NW == 8 ? CDX_GEMM_W8_BK : (WT == 2 ? 16 : 32);          // K tile
    // Global -> LDS staging map (round 4): a wave's load covers FEW rows in FULL 64 / 128-byte runs -- thread -> (row r0 + i * RPI, k
    // quad q): 16 rows x 64 B per load instruction (128 x 128 tile), 8 rows x 128 B (64 x 64) -- instead of 64 rows x 16 B: a quarter /
    // an eighth of the cache lines per instruction through the CU's memory pipe, which is what the co-resident workgroups' epilogue
    // stores queue behind (profiles/r04_gemm_direct_epilogue_ab.txt).  LD: the transposed ds_write_b32 of a 32-lane half hit 32 banks.
    constexpr int QPR = BK / 4;                    // float4 quads per row and K tile
    constexpr int RPI = THREADS / QPR;             // rows one load of the whole workgroup covers
    constexpr int NLD = BMN / RPI;                 // loads per operand and thread (2; 1 in the 8-wave shape)
    static_assert(NLD == 2 || (NW == 8 && NLD == 1), "two float4 per operand and thread (one with 8 waves)");
    constexpr int LD = WT == 2 ? BMN + 2 : BMN + 1;
    // one LDS arena: two stages of A/B staging tiles [k][row] during the K loop, then NW wave-private 32 x 36 transposition patches
    constexpr int SMEM_FLOATS = (4 * BK * LD > NW * 32 * GM_EP_LD) ? 4 * BK * LD : NW * 32 * GM_EP_LD;
#if CDX_GEMM_W8_BK == 32
    // (66.5 KB for the 8-wave shape with 32-wide K tiles: past the 64 KB of static LDS -- dynamic, requested by the launcher)
    extern __shared__ __attribute__((aligned(16))) float dyn_smem[];
    __shared__ __attribute__((aligned(16))) float sta_smem[NW == 8 ? 4 : SMEM_FLOATS];
    float* smem = NW == 8 ? dyn_smem : sta_smem;
#else
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
#endif
    float (*As)[LD] = reinterpret_cast<float (*)[LD]>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (g.M + BMN - 1) / BMN, tiles_n = (g.N + BMN - 1) / BMN;
    const int n_tiles = tiles_m * tiles_n;
    constexpr bool PERSIST = CDX_GEMM_PERSIST && NW == 8 && FAST;
    const int vb_end = PERSIST ? n_tiles * k_split : (int)blockIdx.x + 1;
    bool prefetched = false;
    float4 ra[NLD], rb[NLD];
  for (int vb = blockIdx.x; vb < vb_end; vb += (PERSIST ? (int)gridDim.x : 1 << 30)) {
    const int lin = vb % n_tiles, slice = vb / n_tiles;     // split-K: slice of the K range
    // Tiles are walked n-fastest: concurrently resident workgroups share a few A row blocks across all their N tiles (the whole W
    // fits L2), instead of re-fetching each A block N/128 times -- measured +15-17 % on the config-4/5 shapes over m-fastest.
    // Optional XCD-aware variant (workgroup b runs on XCD b % 8): one contiguous range of that list per XCD.
    int tile = lin;
    if (xcd_order) {
        const int xcd = lin & 7, t = lin >> 3, q8 = n_tiles >> 3, r8 = n_tiles & 7;
        tile = xcd * q8 + min(xcd, r8) + t;
    }
    const int bm = (tile / tiles_n) * BMN, bn = (tile % tiles_n) * BMN;
    const int q4 = (tid % QPR) * 4, r0 = tid / QPR;     // this thread stages k quad q4 / 4 of rows r0 and r0 + RPI

    gm_stamp(0);
    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int lrow[NLD], arow[NLD], wrow[NLD];
    const float *ap[NLD], *wp[NLD];
    // implicit-GEMM conv (FAST: conv_cin % 4 == 0, so an aligned float4 never straddles two taps)
    constexpr bool conv = CONV;                      // compile-time: the plain-GEMM instantiations carry no conv code at all
    int conv_in0[NLD];
    size_t conv_base[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        lrow[i] = r0 + i * RPI;
        arow[i] = FAST ? min(bm + lrow[i], g.M - 1) : bm + lrow[i];
        wrow[i] = FAST ? min(bn + lrow[i], g.N - 1) : bn + lrow[i];
        ap[i] = g.A + (size_t)arow[i] * g.lda + q4;
        wp[i] = g.W + (size_t)wrow[i] * g.ldw + q4;
        conv_in0[i] = 0; conv_base[i] = 0;
        if (conv) {
            const int cb = arow[i] / g.conv_lout, clo = arow[i] - cb * g.conv_lout;
            conv_in0[i] = clo * g.conv_stride - g.conv_pad;
            conv_base[i] = (size_t)cb * g.conv_lin;
        }
    }
    const float conv_inv_cin = conv ? 1.0f / (float)g.conv_cin : 0.f;
    // The validity of a conv tap is applied when the registers are parked in LDS, NOT at load time: a select right behind the load
    // makes hipcc wait for it (s_waitcnt vmcnt(0) in the middle of the MFMA stream: -5..-17 % measured on the large GEMMs).
    bool cok[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) cok[i] = true;
    auto fetch = [&](int kt) {                           // global -> registers: k quad q4 of this thread's two rows
        const int k = kt + q4;
        if (FAST) {
            int tap = 0, c0 = 0;
            if (conv) {                                  // 4 consecutive channels of one tap (clamped inside the tensor)
                tap = (int)(((float)k + 0.5f) * conv_inv_cin);
                c0 = k - tap * g.conv_cin;
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const float* pa = ap[i] + kt;
                if (conv) {
                    const int pos = conv_in0[i] + tap;
                    cok[i] = pos >= 0 && pos < g.conv_lin;
                    pa = g.A + (conv_base[i] + (cok[i] ? pos : 0)) * g.lda + c0;
                }
                ra[i] = *reinterpret_cast<const float4*>(pa);
                rb[i] = *reinterpret_cast<const float4*>(wp[i] + kt);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                ra[i] = conv ? gm_conv_load4(g, arow[i], k) : gm_load4_guarded(g.A, arow[i], g.M, k, g.K, g.lda);
                rb[i] = gm_load4_guarded(g.W, wrow[i], g.N, k, g.K, g.ldw);
            }
        }
    };
    auto stage = [&](int buf) {                          // registers -> LDS stage `buf`, transposed to [k][row]
        float (*Ad)[LD] = As + buf * (2 * BK);
        float (*Bd)[LD] = Ad + BK;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (FAST && conv && !cok[i]) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            Ad[q4 + 0][lrow[i]] = ra[i].x; Ad[q4 + 1][lrow[i]] = ra[i].y; Ad[q4 + 2][lrow[i]] = ra[i].z; Ad[q4 + 3][lrow[i]] = ra[i].w;
            Bd[q4 + 0][lrow[i]] = rb[i].x; Bd[q4 + 1][lrow[i]] = rb[i].y; Bd[q4 + 2][lrow[i]] = rb[i].z; Bd[q4 + 3][lrow[i]] = rb[i].w;
        }
    };

    const int wm = NW == 8 ? (wave >> 2) * 64 : (wave >> 1) * (32 * WT), wn = NW == 8 ? (wave & 3) * 32 : (wave & 1) * (32 * WT);
    const int lr = lane & 31, lk = lane >> 5;
    const int nk_all = (g.K + BK - 1) / BK;
    const int per = (nk_all + k_split - 1) / k_split;    // K tiles per slice
    const int kt0 = slice * per * BK;                     // first k of this slice
    const int nk = min(per, nk_all - slice * per);

    // Two LDS stages, ONE barrier per K tile: while the MFMAs of tile t run out of stage t & 1, the registers
    // holding tile t + 1 (fetched a whole tile earlier) are written to the other stage and tile t + 2 is requested.
    if (!(PERSIST && prefetched)) fetch(kt0);            // (persistent walk: requested before the previous tile's epilogue)
    stage(0);
    if (nk > 1) fetch(kt0 + BK);
    __syncthreads();
    gm_stamp(1);
#if CDX_GEMM_KBLOCK
    f32x16 blk[WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) blk[i][r] = 0.f;
#endif
#if CDX_GEMM_KBLOCK
    if constexpr (NW == 8) {
        // 8-wave shape: two row blocks x one column block per wave = two independent chains; the block accumulators run over TWO K
        // tiles (the second tile continues the chains of the first), then both are flushed into the totals
        auto tile8 = [&](int t, auto first) {
            const float (*Ac)[LD] = As + (t & 1) * (2 * BK);
            const float (*Bc)[LD] = Ac + BK;
            float av[2] = {Ac[lk][wm + lr], Ac[lk][wm + 32 + lr]}, bv = Bc[lk][wn + lr];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float na[2] = {0.f, 0.f}, nb = 0.f;
                if (kk + 2 < BK) { na[0] = Ac[kk + 2 + lk][wm + lr]; na[1] = Ac[kk + 2 + lk][wm + 32 + lr]; nb = Bc[kk + 2 + lk][wn + lr]; }
                if (decltype(first)::value && kk == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) blk[i][r] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) blk[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, blk[i], 0, 0, 0);
                if (kk == BK / 2 - 2 && t + 1 < nk) {        // mid-tile: park tile t + 1 in the other stage, request tile t + 2
                    stage((t + 1) & 1);
                    if (t + 2 < nk) fetch(kt0 + (t + 2) * BK);
                }
                av[0] = na[0]; av[1] = na[1]; bv = nb;
            }
            __syncthreads();                             // stage (t+1)&1 complete, stage t&1 free for tile t + 2
        };
        for (int t = 0; t < nk; t += (BK == 32 ? 1 : 2)) {
            tile8(t, std::true_type{});
            if (BK != 32 && t + 1 < nk) tile8(t + 1, std::false_type{});
#pragma unroll
            for (int i = 0; i < 2; ++i) gm_flush(acc[i][0], blk[i]);
        }
    } else
#else
    static_assert(NW == 4, "the 8-wave shape exists for the K-blocked accumulation");
#endif
    for (int t = 0; t < nk; ++t) {
        const float (*Ac)[LD] = As + (t & 1) * (2 * BK);
        const float (*Bc)[LD] = Ac + BK;
#if CDX_GEMM_KBLOCK
        // K-BLOCKED accumulation (round 5): the products of ONE K tile are summed in block accumulators that start at zero, and the block
        // sums are added to the running totals -- an element's rounding error grows with sqrt(K / BK) + sqrt(BK) roundings instead of
        // the sqrt(K) of one sequential fma chain (profiles/r05_dit_error_budget.txt: K = 320 3.1e-6 -> 0.8e-6 of the output's rms,
        // K = 1280 with gate / residual 9.6e-7 -> 2.5e-7; ATen's CPU kernels on the same inputs: 2.0e-6 / 2.7e-7).  Registers: the
        // wave's rows are walked in WT halves so that only WT block accumulators (32 registers at WT = 2) live next to the 64 of the
        // total (154 VGPRs as before, three workgroups per CU).  In THIS 4-wave shape the flush costs 3-6 % against the sequential chain
        // (profiles/r05_gemm_kblock_ab.txt): s_nop for the last MFMA + 16 v_pk_add_f32 per half are issue time the wave's MFMAs do not
        // get, and three waves per SIMD do not cover it.  Hiding it inside this shape was tried twice and dropped (single-chain segments:
        // -4.5..-6.6 %; a software-pipelined flush: hipcc renames the restarted accumulator, 40-50 spilled VGPRs); the 8-wave shape above
        // is what won the cost back (0..-2 %, profiles/r05_gemm_w8_ab.txt).  This path serves the guarded-load (!FAST) launches and
        // CDX_GEMM_W8=0, and -- WT = 1 -- the 64 x 64 tiles.
#pragma unroll
        for (int h = 0; h < WT; ++h) {
#pragma unroll
            for (int j = 0; j < WT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) blk[j][r] = 0.f;
            float av = Ac[lk][wm + 32 * h + lr], bv[WT];
#pragma unroll
            for (int j = 0; j < WT; ++j) bv[j] = Bc[lk][wn + 32 * j + lr];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float na = 0.f, nb[WT];
#pragma unroll
                for (int j = 0; j < WT; ++j) nb[j] = 0.f;
                if (kk + 2 < BK) {
                    na = Ac[kk + 2 + lk][wm + 32 * h + lr];
#pragma unroll
                    for (int j = 0; j < WT; ++j) nb[j] = Bc[kk + 2 + lk][wn + 32 * j + lr];
                }
#pragma unroll
                for (int j = 0; j < WT; ++j) blk[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], blk[j], 0, 0, 0);
                if (h == 0 && kk == (WT == 2 ? GM_STAGE_AT : BK / 2 - 4) && t + 1 < nk) {   // park tile t + 1 in the other stage, request t + 2
                    stage((t + 1) & 1);
                    if (t + 2 < nk) fetch(kt0 + (t + 2) * BK);
                }
                av = na;
#pragma unroll
                for (int j = 0; j < WT; ++j) bv[j] = nb[j];
            }
#pragma unroll
            for (int j = 0; j < WT; ++j) gm_flush(acc[h][j], blk[j]);
        }
#else
        // operands of the next k pair are read from LDS before the MFMAs of the current one are issued
        float av[WT], bv[WT];
#pragma unroll
        for (int i = 0; i < WT; ++i) { av[i] = Ac[lk][wm + 32 * i + lr]; bv[i] = Bc[lk][wn + 32 * i + lr]; }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float na[WT], nb[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                na[i] = 0.f; nb[i] = 0.f;
                if (kk + 2 < BK) { na[i] = Ac[kk + 2 + lk][wm + 32 * i + lr]; nb[i] = Bc[kk + 2 + lk][wn + 32 * i + lr]; }
            }
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            if (kk == BK / 2 - 4 && t + 1 < nk) {        // mid-tile: park tile t + 1 in the other stage, request tile t + 2
                stage((t + 1) & 1);
                if (t + 2 < nk) fetch(kt0 + (t + 2) * BK);
            }
#pragma unroll
            for (int i = 0; i < WT; ++i) { av[i] = na[i]; bv[i] = nb[i]; }
        }
#endif
        __syncthreads();                                 // stage (t+1)&1 complete, stage t&1 free for tile t + 2
    }

    asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[WTM - 1][WTN - 1][15]));
    gm_stamp(2);
    // (the loop's last barrier already separates the final LDS reads from the patches written below)
    // fused epilogue, one specialisation per activation (the branch is uniform; only the taken copy touches the I-cache)
    const int row0 = bm + wm, col0 = bn + wn;
    if constexpr (PERSIST) {
        // the NEXT tile of this workgroup: its first K tile is requested now and lands during the epilogue below (plain GEMMs only: a
        // conv's tap validity belongs to the tile's own set-up)
        prefetched = false;
        const int vn = vb + (int)gridDim.x;
        if (vn < vb_end && !conv) {
            const int ln = vn % n_tiles, sn = vn / n_tiles;
            int tn = ln;
            if (xcd_order) {
                const int xcd = ln & 7, t = ln >> 3, q8 = n_tiles >> 3, r8 = n_tiles & 7;
                tn = xcd * q8 + min(xcd, r8) + t;
            }
            const int bmn_ = (tn / tiles_n) * BMN, bnn_ = (tn % tiles_n) * BMN;
            const int ktn = sn * per * BK;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                ra[i] = *reinterpret_cast<const float4*>(g.A + (size_t)min(bmn_ + lrow[i], g.M - 1) * g.lda + q4 + ktn);
                rb[i] = *reinterpret_cast<const float4*>(g.W + (size_t)min(bnn_ + lrow[i], g.N - 1) * g.ldw + q4 + ktn);
            }
            prefetched = true;
        }
    }
    float* patch = smem + wave * (32 * GM_EP_LD);
    const bool fe = fast_ep != 0;
    if (k_split > 1) {                                   // raw partial tile; the epilogue runs in gm_splitk_reduce_kernel
        cdx_gemm_args gp = g;
        gp.C = g.partial + (size_t)slice * g.M * g.N; gp.ldc = g.N;
        gp.bias = nullptr; gp.gate = nullptr; gp.residual = nullptr; gp.table = nullptr;
        gm_epilogue_any<CDX_ACT_NONE, WTM, WTN>(gp, acc, patch, row0, col0, lane, (g.N % 4 == 0));
        gm_stamp(3);
        if (PERSIST) { __syncthreads(); continue; }
        return;
    }
    switch (g.act) {
        case CDX_ACT_MISH: gm_epilogue_any<CDX_ACT_MISH, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
        case CDX_ACT_GELU_ERF: gm_epilogue_any<CDX_ACT_GELU_ERF, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
        case CDX_ACT_LEAKY: gm_epilogue_any<CDX_ACT_LEAKY, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
        case CDX_ACT_SILU: gm_epilogue_any<CDX_ACT_SILU, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
        case CDX_ACT_RELU: gm_epilogue_any<CDX_ACT_RELU, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
        case CDX_ACT_GELU_TANH: gm_epilogue_any<CDX_ACT_GELU_TANH, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
        case CDX_ACT_TANH: gm_epilogue_any<CDX_ACT_TANH, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
        default: gm_epilogue_any<CDX_ACT_NONE, WTM, WTN>(g, acc, patch, row0, col0, lane, fe); break;
    }
    gm_stamp(3);
    if (PERSIST) __syncthreads();                        // the patches alias the staging area of the next tile
  }
}

// ------------------------------------------------------------------------------------------------
// Second pass of a split-K launch: C = epilogue(sum over slices, in slice order, of partial[slice]) -- same epilogue semantics as
// the GEMM kernel (bias -> act -> gate -> residual -> table), one output element per thread, memory bound.
// ------------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(256) void gm_splitk_reduce_kernel(const cdx_gemm_args g, const int k_split) {
    constexpr int W = VEC4 ? 4 : 1;                      // VEC4: N, ldc, ldr, ldg multiples of 4 and 16-byte aligned bases
    const size_t total = (size_t)g.M * g.N, groups = total / W;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < groups; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i = q * W;
        const int m = (int)(i / g.N), n = (int)(i - (size_t)m * g.N);
        float v[W];
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = 0.f;
        for (int s = 0; s < k_split; ++s) {
            const float* p = g.partial + (size_t)s * total + i;
            if (VEC4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] += t.x; v[1 % W] += t.y; v[2 % W] += t.z; v[3 % W] += t.w; }
            else v[0] += p[0];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            float x = gm_act(v[j] + (g.bias ? g.bias[n + j] : 0.f), g.act);
            if (g.gate) x *= g.gate[(size_t)(m / g.rows_per_gate) * g.ldg + n + j];
            if (g.residual) x += g.residual[(size_t)m * g.ldr + n + j];
            if (g.table) x += g.table[(size_t)(m % g.table_rows) * g.N + n + j];
            v[j] = x;
        }
        float* c = g.C + (size_t)m * g.ldc + n;
        if (VEC4) *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1 % W], v[2 % W], v[3 % W]);
        else c[0] = v[0];
    }
}

// ------------------------------------------------------------------------------------------------
// Row LayerNorm (no affine, eps) + adaLN modulation:  y[m][c] = LN(x[m])[c] * (1 + scale[b][c]) + shift[b][c]
// (reference dit.py:10-11, 33-35, 48); b = m / rows_per_mod.  With gamma/beta instead of shift/scale it is a plain
// affine LayerNorm (idqlmlp.py:14).  One wave per row, values in registers, DPP-free shuffles (memory-bound op).
// ------------------------------------------------------------------------------------------------
// NT = values per lane: 16 (C <= 1024), 32 (C <= 2048) or 64 (C <= 4096; the wide IDQLMlp variants)
template <int NT>
__global__ __launch_bounds__(256) void cdx_layernorm_kernel(const cdx_ln_args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const float* x = a.x + (size_t)(a.x_rows > 0 ? row % a.x_rows : row) * a.ldx;
    float v[NT];                                        // C <= 64 * NT
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
    float* y = a.y + (size_t)row * a.ldy;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = lane + 64 * t;
        if (c < a.C) {
            float o = v[t] * rstd;
            if (a.gamma) o = o * a.gamma[c] + a.beta[c];
            if (a.scale) o = o * (1.0f + a.scale[(size_t)b * a.ldmod + c]) + a.shift[(size_t)b * a.ldmod + c];
            y[c] = o;
        }
    }
}

// The same LayerNorm with 16-byte accesses: a row belongs to 16 lanes (NV float4 each: C <= 64 NV), four rows per wave, sixteen per
// workgroup.  The scalar kernel above keeps ONE 4-byte load per lane and 64 columns in flight -- 1.8 TB/s on the (32 768 x 320) token
// rows of config 4 (profiles/r05_cfg4_512_rocprofv3_kernel_stats.csv: 47 us per call, 6.5 % of the sampling loop); here a lane has up
// to NV independent 16-byte loads outstanding.  Same two-pass statistics (mean, then centred squares); the order of the sums differs
// from the scalar kernel's (results agree to rounding).  Needs C % 4 == 0 and 16-byte aligned rows / parameter vectors (the host checks).
template <int NV>
__global__ __launch_bounds__(256) void cdx_layernorm_vec_kernel(const cdx_ln_args a) {
    const int sub = threadIdx.x & 15;
    const int row0 = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row0 < a.M;
    const int row = live ? row0 : a.M - 1;               // (idle lanes of the last workgroup recompute the last row and store nothing)
    const float* x = a.x + (size_t)(a.x_rows > 0 ? row % a.x_rows : row) * a.ldx;
    const int n4 = a.C >> 2;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int q = sub + 16 * t;
        v[t] = q < n4 ? *reinterpret_cast<const float4*>(x + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)a.C;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        if (sub + 16 * t < n4) {
            v[t].x -= mean; v[t].y -= mean; v[t].z -= mean; v[t].w -= mean;
            s2 += (v[t].x * v[t].x + v[t].y * v[t].y) + (v[t].z * v[t].z + v[t].w * v[t].w);
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = 1.0f / sqrtf(s2 / (float)a.C + a.eps);
    if (!live) return;
    const size_t mod = (size_t)(row / a.rows_per_mod) * a.ldmod;
    float* y = a.y + (size_t)row * a.ldy;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int q = sub + 16 * t;
        if (q < n4) {
            float4 o = make_float4(v[t].x * rstd, v[t].y * rstd, v[t].z * rstd, v[t].w * rstd);
            if (a.gamma) {
                const float4 g = *reinterpret_cast<const float4*>(a.gamma + 4 * q), be = *reinterpret_cast<const float4*>(a.beta + 4 * q);
                o = make_float4(o.x * g.x + be.x, o.y * g.y + be.y, o.z * g.z + be.z, o.w * g.w + be.w);
            }
            if (a.scale) {
                const float4 sc = *reinterpret_cast<const float4*>(a.scale + mod + 4 * q), sh = *reinterpret_cast<const float4*>(a.shift + mod + 4 * q);
                o = make_float4(o.x * (1.0f + sc.x) + sh.x, o.y * (1.0f + sc.y) + sh.y, o.z * (1.0f + sc.z) + sh.z, o.w * (1.0f + sc.w) + sh.w);
            }
            *reinterpret_cast<float4*>(y + 4 * q) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (+ activation, FiLM, residual) on channel-last rows: one wave per (sample, group); the group's L x C/G values stay
// in registers when there are <= 2048 of them (two-pass variance like ATen), else they are re-read.
// ------------------------------------------------------------------------------------------------
#define GN_REGS 32
__global__ __launch_bounds__(256) void cdx_groupnorm_kernel(const cdx_gn_args a) {
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wg >= a.B * a.G) return;
    const int b = wg / a.G, grp = wg - b * a.G;
    const int cg = a.C / a.G, n = a.L * cg;
    const float inv_cg = 1.0f / (float)cg;
    const float* xb = a.x + (size_t)b * a.L * a.ldx + grp * cg;
    const bool in_regs = n <= 64 * GN_REGS;
    float v[GN_REGS];
    float s = 0.f;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < GN_REGS; ++i) {
            const int e = lane + 64 * i;
            float x = 0.f;
            if (e < n) {
                const int l = (int)(((float)e + 0.5f) * inv_cg), c = e - l * cg;
                x = xb[(size_t)l * a.ldx + c];
            }
            v[i] = x;
            s += x;
        }
    } else {
        for (int e = lane; e < n; e += 64) {
            const int l = e / cg, c = e - l * cg;
            s += xb[(size_t)l * a.ldx + c];
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)n;
    float s2 = 0.f;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < GN_REGS; ++i) {
            const float d = (lane + 64 * i < n) ? v[i] - mean : 0.f;
            s2 += d * d;
        }
    } else {
        for (int e = lane; e < n; e += 64) {
            const int l = e / cg, c = e - l * cg;
            const float d = xb[(size_t)l * a.ldx + c] - mean;
            s2 += d * d;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = 1.0f / sqrtf(s2 / (float)n + a.eps);
    const float* fa = a.fa ? a.fa + (size_t)(a.fa_per_sample ? b : a.fa_row) * a.ldfa : nullptr;
    const float* fb = a.fb ? a.fb + (size_t)b * a.ldfb : nullptr;
    auto emit = [&](int e, float x) {
        const int l = (int)(((float)e + 0.5f) * inv_cg), c = e - l * cg, ch = grp * cg + c;
        float y = gm_act((x - mean) * rstd * a.gamma[ch] + a.beta[ch], a.act);
        if (a.film_mode == 1) {
            const float sc = (fa ? fa[ch] : 0.f) + (fb ? fb[ch] : 0.f);
            const float bi = (fa ? fa[a.C + ch] : 0.f) + (fb ? fb[a.C + ch] : 0.f);
            y = sc * y + bi;
        } else if (a.film_mode == 2) {
            y += (fa ? fa[ch] : 0.f) + (fb ? fb[ch] : 0.f);
        }
        const size_t row = (size_t)b * a.L + l;
        if (a.residual) y += a.residual[row * a.ldr + ch];
        a.y[row * a.ldy + ch] = y;
    };
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < GN_REGS; ++i) {
            const int e = lane + 64 * i;
            if (e < n) emit(e, v[i]);
        }
    } else {
        for (int e = lane; e < n; e += 64) {
            const int l = e / cg, c = e - l * cg;
            emit(e, xb[(size_t)l * a.ldx + c]);
        }
    }
}

// float4 variant: C/G, every leading dimension and the FiLM strides multiples of 4, 16-byte aligned bases, the group's values in
// registers (L * C/G <= 2048).  A lane owns 4 consecutive channels of a position: x, gamma, beta, the FiLM rows, the residual and y move
// as dwordx4 (the scalar kernel issues ~8 dword loads per element: measured 1.9 TB/s on the config-3 tensors).  Mean, then the centred
// sum of squares, accumulated in float64.
// SLICES: x is not one tensor but the raw K-slice partial sums a split-K conv left behind (x = slice 0, the others `slice_stride`
// floats apart, `slices` of them) plus the conv's bias: the kernel sums them in slice order, adds the bias -- the float operations of
// gm_splitk_reduce_kernel, in its order -- and normalises; the conv's second pass and the round trip of its output are gone
// (config 3: 36 of them per forward).
#define GN_VREGS 8
template <bool SLICES>
__global__ __launch_bounds__(256) void cdx_groupnorm_vec_kernel(const cdx_gn_args a, const int slices, const size_t slice_stride,
                                                                const float* __restrict__ xbias) {
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wg >= a.B * a.G) return;
    const int b = wg / a.G, grp = wg - b * a.G;
    const int cg = a.C / a.G, cq = cg >> 2, n4 = a.L * cq, n = a.L * cg;
    const float inv_cq = 1.0f / (float)cq;
    const float* xb = a.x + (size_t)b * a.L * a.ldx + grp * cg;
    float4 v[GN_VREGS];
    int l_of[GN_VREGS], c_of[GN_VREGS];
    double s = 0.0;        // statistics in float64: the kernel is memory-bound, and the executor's deviation from the reference then is
                           // the reference's own fp32 rounding alone (a clipped eps-prediction loop amplifies every ulp of it)
#pragma unroll
    for (int i = 0; i < GN_VREGS; ++i) {
        const int e = lane + 64 * i;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        int l = 0, c = 0;
        if (e < n4) {
            l = (int)(((float)e + 0.5f) * inv_cq);
            c = 4 * (e - l * cq);
            const float* px = xb + (size_t)l * a.ldx + c;
            x = *reinterpret_cast<const float4*>(px);
            if (SLICES) {
                for (int sl = 1; sl < slices; ++sl) {
                    const float4 t = *reinterpret_cast<const float4*>(px + (size_t)sl * slice_stride);
                    x.x += t.x; x.y += t.y; x.z += t.z; x.w += t.w;
                }
                if (xbias) {
                    const float4 t = *reinterpret_cast<const float4*>(xbias + grp * cg + c);
                    x.x += t.x; x.y += t.y; x.z += t.z; x.w += t.w;
                }
            }
        }
        v[i] = x; l_of[i] = l; c_of[i] = c;
        s += ((double)x.x + (double)x.y) + ((double)x.z + (double)x.w);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const double mean_d = s / (double)n;
    const float mean = (float)mean_d;
    double s2 = 0.0;
#pragma unroll
    for (int i = 0; i < GN_VREGS; ++i) {
        if (lane + 64 * i < n4) {
            const double d0 = (double)v[i].x - mean_d, d1 = (double)v[i].y - mean_d, d2 = (double)v[i].z - mean_d, d3 = (double)v[i].w - mean_d;
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = (float)(1.0 / sqrt(s2 / (double)n + (double)a.eps));
    const float* fa = a.fa ? a.fa + (size_t)(a.fa_per_sample ? b : a.fa_row) * a.ldfa : nullptr;
    const float* fb = a.fb ? a.fb + (size_t)b * a.ldfb : nullptr;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
#pragma unroll
    for (int i = 0; i < GN_VREGS; ++i) {
        if (lane + 64 * i >= n4) continue;
        const int ch = grp * cg + c_of[i];
        const float4 ga = ld4(a.gamma + ch), be = ld4(a.beta + ch);
        float y[4] = {gm_act((v[i].x - mean) * rstd * ga.x + be.x, a.act), gm_act((v[i].y - mean) * rstd * ga.y + be.y, a.act),
                      gm_act((v[i].z - mean) * rstd * ga.z + be.z, a.act), gm_act((v[i].w - mean) * rstd * ga.w + be.w, a.act)};
        if (a.film_mode != 0) {
            float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), bi = sc;
            const int boff = a.film_mode == 1 ? a.C : 0;              // mode 1: [scale (C) | bias (C)], mode 2: [bias (C)]
            if (fa) { if (a.film_mode == 1) sc = ld4(fa + ch); bi = ld4(fa + boff + ch); }
            if (fb) {
                const float4 t = ld4(fb + boff + ch);
                bi.x += t.x; bi.y += t.y; bi.z += t.z; bi.w += t.w;
                if (a.film_mode == 1) { const float4 u = ld4(fb + ch); sc.x += u.x; sc.y += u.y; sc.z += u.z; sc.w += u.w; }
            }
            if (a.film_mode == 1) { y[0] = sc.x * y[0] + bi.x; y[1] = sc.y * y[1] + bi.y; y[2] = sc.z * y[2] + bi.z; y[3] = sc.w * y[3] + bi.w; }
            else { y[0] += bi.x; y[1] += bi.y; y[2] += bi.z; y[3] += bi.w; }
        }
        const size_t row = (size_t)b * a.L + l_of[i];
        if (a.residual) {
            const float4 r = ld4(a.residual + row * a.ldr + ch);
            y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
        }
        *reinterpret_cast<float4*>(a.y + row * a.ldy + ch) = make_float4(y[0], y[1], y[2], y[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of act(GroupNorm(x) gamma + beta) w.r.t. x, one wave per (sample, group):
//   dz = dy * act'(z),  g = dz * gamma,  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))      (means over the group)
// ------------------------------------------------------------------------------------------------
// SUMS (training with cdx_gn_args.dgamma_sum / dbeta_sum): the four waves of a workgroup take FOUR SAMPLES OF ONE GROUP, combine their
// per-channel sums in LDS and issue one float atomic per channel and workgroup -- B / 4 instead of B adds per address (with one atomic
// per sample the backward of a config-2 step spent 507 us in this kernel against 254 us without the sums, profiles/r05_update_census.txt).
template <bool SUMS>
__global__ __launch_bounds__(256) void cdx_groupnorm_bwd_kernel(const cdx_gn_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float red[SUMS ? 2 * 4 * 256 : 1];
    int b, grp;
    if (SUMS) {
        grp = blockIdx.x % a.G;
        b = (blockIdx.x / a.G) * 4 + wave;
    } else {
        const int wg = blockIdx.x * 4 + wave;
        b = wg / a.G;
        grp = wg - b * a.G;
    }
    const bool live = b < a.B;
    if (!SUMS && !live) return;
    if (SUMS && !live) b = a.B - 1;                       // (an idle wave of the last workgroup recomputes the last sample, adds and stores nothing)
    const int cg = a.C / a.G, n = a.L * cg;
    const float* xb = a.x + (size_t)b * a.L * a.ldx + grp * cg;
    const float* db = a.residual + (size_t)b * a.L * a.ldr + grp * cg;
    float s = 0.f;
    for (int e = lane; e < n; e += 64) {
        const int l = e / cg, c = e - l * cg;
        s += xb[(size_t)l * a.ldx + c];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)n;
    float s2 = 0.f;
    for (int e = lane; e < n; e += 64) {
        const int l = e / cg, c = e - l * cg;
        const float d = xb[(size_t)l * a.ldx + c] - mean;
        s2 += d * d;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = 1.0f / sqrtf(s2 / (float)n + a.eps);
    float sg = 0.f, sgx = 0.f;                           // sum g, sum g * xhat
    // this lane's channels: sum dz * xhat, sum dz (training: d gamma, d beta).  cg is a power of two <= 256 (checked by the host entry):
    // up to 64 a lane's channel e % cg is the same in every iteration; 128 / 256 (ChiUNet1d at 1024 / 2048 channels in 8 groups): a lane
    // walks the channels lane + 64 j, j = iteration % (cg / 64) -- one partial pair per j
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f}, pd[4] = {0.f, 0.f, 0.f, 0.f};
    const int jmask = cg > 64 ? cg / 64 - 1 : 0;
    int it = 0;
    for (int e = lane; e < n; e += 64, ++it) {
        const int l = e / cg, c = e - l * cg, ch = grp * cg + c;
        const float xh = (xb[(size_t)l * a.ldx + c] - mean) * rstd;
        const float z = xh * a.gamma[ch] + a.beta[ch];
        const float dyv = db[(size_t)l * a.ldr + c];
        const float dz = dyv * (a.act == CDX_ACT_MISH ? gm_act(z, CDX_ACT_MISH_GRAD) : 1.0f);
        const float g = dz * a.gamma[ch];
        sg += g;
        sgx += g * xh;
        const int j = it & jmask;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q == j) { pg[q] += dz * xh; pb[q] += dz; pd[q] += dyv; }
    }
    if (a.dy_possum != nullptr) {                        // per sample and channel: the sum over the positions of d loss / d y
        if (cg <= 64) {
            for (int o = 32; o >= cg; o >>= 1) pd[0] += __shfl_xor(pd[0], o, 64);
            if (lane < cg && live) a.dy_possum[(size_t)b * a.ld_possum + grp * cg + lane] = pd[0];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q <= jmask && live) a.dy_possum[(size_t)b * a.ld_possum + grp * cg + lane + 64 * q] = pd[q];
        }
    }
    if (SUMS || a.dgamma_part != nullptr) {
        // per-sample partials (B, C) for the caller's column sums, or -- SUMS -- this workgroup's four samples combined in LDS
        float* rg = SUMS ? red + wave * 256 : red;
        float* rb = SUMS ? red + (4 + wave) * 256 : red;
        if (cg <= 64) {       // the lanes that share a channel differ in the bits >= log2(cg)
            for (int o = 32; o >= cg; o >>= 1) { pg[0] += __shfl_xor(pg[0], o, 64); pb[0] += __shfl_xor(pb[0], o, 64); }
            if (lane < cg) {
                const int ch = grp * cg + lane;
                if (!SUMS) { a.dgamma_part[(size_t)b * a.C + ch] = pg[0]; a.dbeta_part[(size_t)b * a.C + ch] = pb[0]; }
                else { rg[lane] = live ? pg[0] : 0.f; rb[lane] = live ? pb[0] : 0.f; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q <= jmask) {
                    const int c = lane + 64 * q, ch = grp * cg + c;
                    if (!SUMS) { a.dgamma_part[(size_t)b * a.C + ch] = pg[q]; a.dbeta_part[(size_t)b * a.C + ch] = pb[q]; }
                    else { rg[c] = live ? pg[q] : 0.f; rb[c] = live ? pb[q] : 0.f; }
                }
        }
        if (SUMS) {
            __syncthreads();
            for (int c = threadIdx.x; c < cg; c += 256) {
                atomicAdd(a.dgamma_sum + grp * cg + c, (red[c] + red[256 + c]) + (red[512 + c] + red[768 + c]));
                atomicAdd(a.dbeta_sum + grp * cg + c, (red[1024 + c] + red[1280 + c]) + (red[1536 + c] + red[1792 + c]));
            }
            if (!live) return;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
    const float mg = sg / (float)n, mgx = sgx / (float)n;
    for (int e = lane; e < n; e += 64) {
        const int l = e / cg, c = e - l * cg, ch = grp * cg + c;
        const float xh = (xb[(size_t)l * a.ldx + c] - mean) * rstd;
        const float z = xh * a.gamma[ch] + a.beta[ch];
        const float dz = db[(size_t)l * a.ldr + c] * (a.act == CDX_ACT_MISH ? gm_act(z, CDX_ACT_MISH_GRAD) : 1.0f);
        const float g = dz * a.gamma[ch];
        a.y[((size_t)b * a.L + l) * a.ldy + ch] = rstd * (g - mg - xh * mgx);
    }
}

// The same backward with the group in REGISTERS (round 6): one wave per (sample, group) as above, but x and dy are read ONCE, as
// dwordx4 (the scalar kernel walks x three times and dy twice with dword loads: 9.3 us per launch on the <= 1 MB tensors of a config-2
// step, 12 % of update()'s device time in 33 launches), statistics in float64 as in the forward kernel.  Needs a power-of-two group
// width of 4..256 channels, at most 8 float4 per lane (L * cg <= 2048) and 16-byte aligned rows.  A lane keeps ONE channel quad
// (64 % (cg / 4) == 0): its items are the positions l0 + i * (256 / cg).
template <bool SUMS>
__global__ __launch_bounds__(256) void cdx_groupnorm_bwd_vec_kernel(const cdx_gn_args a, const int spw) {
    // SUMS: a wave walks `spw` consecutive samples of its group and keeps the per-channel sums in registers across them, so that an
    // address of dgamma_sum / dbeta_sum sees B / (4 spw) float atomics instead of B / 4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float red[SUMS ? 2 * 4 * 256 : 1];
    int b0, grp;
    if (SUMS) {
        grp = blockIdx.x % a.G;
        b0 = ((blockIdx.x / a.G) * 4 + wave) * spw;
    } else {
        const int wg = blockIdx.x * 4 + wave;
        b0 = wg / a.G;
        grp = wg - b0 * a.G;
        if (b0 >= a.B) return;
    }
    const int cg = a.C / a.G, cq = cg >> 2, n = a.L * cg;
    const int sh = __ffs(cq) - 1;
    const int c4 = 4 * (lane & (cq - 1)), l0 = lane >> sh, lstep = 64 >> sh;
    const int ch = grp * cg + c4;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    const float4 ga4 = ld4(a.gamma + ch), be4 = ld4(a.beta + ch);
    const float ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, be[4] = {be4.x, be4.y, be4.z, be4.w};
    const bool sums = SUMS || a.dgamma_part != nullptr;
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int si = 0; si < (SUMS ? spw : 1); ++si) {
        const int b = b0 + si;
        if (b >= a.B) break;                              // (wave-uniform)
        const float* xb = a.x + (size_t)b * a.L * a.ldx + ch;
        const float* db = a.residual + (size_t)b * a.L * a.ldr + ch;
        float4 v[GN_VREGS], d[GN_VREGS];
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < GN_VREGS; ++i) {
            v[i] = d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i * lstep >= a.L) continue;               // (wave-uniform: an item index past the last position is empty in every lane)
            const int l = l0 + i * lstep;
            if (l < a.L) {
                v[i] = ld4(xb + (size_t)l * a.ldx);
                d[i] = ld4(db + (size_t)l * a.ldr);
            }
            s += ((double)v[i].x + (double)v[i].y) + ((double)v[i].z + (double)v[i].w);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        const double mean_d = s / (double)n;
        const float mean = (float)mean_d;
        double s2 = 0.0;
#pragma unroll
        for (int i = 0; i < GN_VREGS; ++i) {
            if (i * lstep >= a.L) continue;
            if (l0 + i * lstep < a.L) {
                const double d0 = (double)v[i].x - mean_d, d1 = (double)v[i].y - mean_d, d2 = (double)v[i].z - mean_d, d3 = (double)v[i].w - mean_d;
                s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = (float)(1.0 / sqrt(s2 / (double)n + (double)a.eps));
        float sg = 0.f, sgx = 0.f;
        float qg[4] = {0.f, 0.f, 0.f, 0.f}, qb[4] = {0.f, 0.f, 0.f, 0.f}, pd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < GN_VREGS; ++i) {
            if (i * lstep >= a.L) continue;
            float xv[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, dv[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[j] - mean) * rstd;
                const float z = xh * ga[j] + be[j];
                const float dz = dv[j] * (a.act == CDX_ACT_MISH ? gm_act(z, CDX_ACT_MISH_GRAD) : 1.0f);
                const float g = dz * ga[j];              // (lanes past the last position carry dy = 0: they add nothing anywhere)
                sg += g;
                sgx += g * xh;
                qg[j] += dz * xh; qb[j] += dz; pd[j] += dv[j];
                xv[j] = xh; dv[j] = g;
            }
            v[i] = make_float4(xv[0], xv[1], xv[2], xv[3]);   // x_hat
            d[i] = make_float4(dv[0], dv[1], dv[2], dv[3]);   // g
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { pg[j] += qg[j]; pb[j] += qb[j]; }
        if (a.dy_possum != nullptr || (!SUMS && sums)) {
            for (int o = 32; o >= cq; o >>= 1) {           // the lanes that share a channel quad differ in the bits >= log2(cq)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!SUMS && sums) { qg[j] += __shfl_xor(qg[j], o, 64); qb[j] += __shfl_xor(qb[j], o, 64); }
                    if (a.dy_possum != nullptr) pd[j] += __shfl_xor(pd[j], o, 64);
                }
            }
            if (lane < cq) {
                if (a.dy_possum != nullptr)
                    *reinterpret_cast<float4*>(a.dy_possum + (size_t)b * a.ld_possum + ch) = make_float4(pd[0], pd[1], pd[2], pd[3]);
                if (!SUMS && sums) {
                    *reinterpret_cast<float4*>(a.dgamma_part + (size_t)b * a.C + ch) = make_float4(qg[0], qg[1], qg[2], qg[3]);
                    *reinterpret_cast<float4*>(a.dbeta_part + (size_t)b * a.C + ch) = make_float4(qb[0], qb[1], qb[2], qb[3]);
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
        const float mg = sg / (float)n, mgx = sgx / (float)n;
#pragma unroll
        for (int i = 0; i < GN_VREGS; ++i) {
            const int l = l0 + i * lstep;
            if (l < a.L)
                *reinterpret_cast<float4*>(a.y + ((size_t)b * a.L + l) * a.ldy + ch) =
                    make_float4(rstd * (d[i].x - mg - v[i].x * mgx), rstd * (d[i].y - mg - v[i].y * mgx),
                                rstd * (d[i].z - mg - v[i].z * mgx), rstd * (d[i].w - mg - v[i].w * mgx));
        }
    }
    if (SUMS) {
        // this wave's samples are summed in registers; combine the lanes of a channel quad, then the four waves in LDS: one float atomic
        // per channel and workgroup
        for (int o = 32; o >= cq; o >>= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { pg[j] += __shfl_xor(pg[j], o, 64); pb[j] += __shfl_xor(pb[j], o, 64); }
        }
        if (lane < cq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[wave * 256 + c4 + j] = pg[j];
                red[(4 + wave) * 256 + c4 + j] = pb[j];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cg; c += 256) {
            atomicAdd(a.dgamma_sum + grp * cg + c, (red[c] + red[256 + c]) + (red[512 + c] + red[768 + c]));
            atomicAdd(a.dbeta_sum + grp * cg + c, (red[1024 + c] + red[1280 + c]) + (red[1536 + c] + red[1792 + c]));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-head self-attention over short sequences (T <= 64 tokens, head_dim <= 64), full softmax, no mask
// (reference dit.py:20,34 = nn.MultiheadAttention(batch_first) core: softmax(q k^T / sqrt(d_h)) v).
// qkv: (B*T, 3*d_model) as produced by in_proj (q | k | v, heads contiguous inside each third).
// One workgroup per (batch, head): K and V of the head in LDS, one thread per query row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cdx_attention_kernel(const cdx_attn_args a) {
    __shared__ float Qs[64][65];
    __shared__ float Ks[64][65];
    __shared__ float Vs[64][65];
    const int t = threadIdx.x;
    const int b = blockIdx.x / a.n_heads, h = blockIdx.x % a.n_heads;
    const int dh = a.head_dim, dm = a.n_heads * a.head_dim;
    const size_t row0 = (size_t)b * a.T;
    for (int i = t; i < 64 * dh; i += 64) {              // rows >= T are zero-filled: 0 * stale-LDS NaN must not leak
        const int tok = i / dh, d = i - tok * dh;
        const bool live = tok < a.T;
        const float* base = a.qkv + (row0 + (live ? tok : 0)) * (size_t)(3 * dm) + h * dh + d;
        Qs[tok][d] = live ? base[0] * a.scale : 0.f;
        Ks[tok][d] = live ? base[dm] : 0.f;
        Vs[tok][d] = live ? base[2 * dm] : 0.f;
    }
    __syncthreads();
    if (t >= a.T) return;
    float p[64];                                         // this query's scores; statically indexed -> registers
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 64; ++j) p[j] = 0.f;
    for (int d = 0; d < dh; ++d) {
        const float qd = Qs[t][d];
#pragma unroll
        for (int j = 0; j < 64; ++j) p[j] = fmaf(qd, Ks[j][d], p[j]);      // Ks[j][d]: same address for all lanes
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        float sc = p[j];
        if (a.mask != nullptr && j < a.T) sc += a.mask[t * a.T + j];
        p[j] = j < a.T ? fmaxf(sc, -3.0e38f) : -3.0e38f;
        mx = fmaxf(mx, p[j]);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        p[j] = j < a.T ? expf(p[j] - mx) : 0.f;
        den += p[j];
    }
    const float inv = 1.0f / den;
    float* op = a.out + (row0 + t) * (size_t)dm + h * dh;
    for (int d = 0; d < dh; ++d) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 64; ++j) o = fmaf(p[j], Vs[j][d], o);
        op[d] = o * inv;
    }
}

// ------------------------------------------------------------------------------------------------
// Longer sequences (64 < T <= CDX_ATTN_MAX_T; DiT1d over horizons no shipped config uses): one wave per (batch, head, block of 64
// queries), one thread per query row, keys / values streamed through LDS in tiles of 64 with an online softmax (running maximum,
// running denominator, head_dim accumulators in registers).  Same math as the kernels below up to the order of the sums.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cdx_attention_long_kernel(const cdx_attn_args a) {
    __shared__ float Ks[64][65];
    __shared__ float Vs[64][65];
    const int t = threadIdx.x;
    const int qblocks = (a.T + 63) / 64;
    const int qb = blockIdx.x % qblocks, bh = blockIdx.x / qblocks;
    const int b = bh / a.n_heads, h = bh % a.n_heads;
    const int dh = a.head_dim, dm = a.n_heads * a.head_dim;
    const size_t row0 = (size_t)b * a.T;
    const int q = qb * 64 + t;
    const bool live_q = q < a.T;
    float qv[64], acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) {
        qv[d] = (live_q && d < dh) ? a.qkv[(row0 + q) * (size_t)(3 * dm) + h * dh + d] * a.scale : 0.f;
        acc[d] = 0.f;
    }
    float mx = -3.0e38f, den = 0.f;
    for (int k0 = 0; k0 < a.T; k0 += 64) {
        __syncthreads();
        for (int i = t; i < 64 * dh; i += 64) {
            const int tok = i / dh, d = i - tok * dh;
            const bool live = k0 + tok < a.T;
            const float* base = a.qkv + (row0 + (live ? k0 + tok : 0)) * (size_t)(3 * dm) + h * dh + d;
            Ks[tok][d] = live ? base[dm] : 0.f;
            Vs[tok][d] = live ? base[2 * dm] : 0.f;
        }
        __syncthreads();
        float p[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) p[j] = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            if (d < dh) {
#pragma unroll
                for (int j = 0; j < 64; ++j) p[j] = fmaf(qv[d], Ks[j][d], p[j]);
            }
        }
        float tile_mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            float sc = p[j];
            const bool live = k0 + j < a.T;
            if (a.mask != nullptr && live && live_q) sc += a.mask[(size_t)q * a.T + k0 + j];
            p[j] = live ? fmaxf(sc, -3.0e38f) : -3.0e38f;
            tile_mx = fmaxf(tile_mx, p[j]);
        }
        const float new_mx = fmaxf(mx, tile_mx);
        const float corr = expf(mx - new_mx);                 // (first tile: exp(-3e38 - m) = 0 on den = 0, acc = 0)
        den *= corr;
#pragma unroll
        for (int d = 0; d < 64; ++d) acc[d] *= corr;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            p[j] = (k0 + j < a.T) ? expf(p[j] - new_mx) : 0.f;
            den += p[j];
        }
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            if (d < dh) {
                float o = acc[d];
#pragma unroll
                for (int j = 0; j < 64; ++j) o = fmaf(p[j], Vs[j][d], o);
                acc[d] = o;
            }
        }
        mx = new_mx;
    }
    if (!live_q) return;
    const float inv = 1.0f / den;
    float* op = a.out + (row0 + q) * (size_t)dm + h * dh;
#pragma unroll
    for (int d = 0; d < 64; ++d)
        if (d < dh) op[d] = acc[d] * inv;
}

// ------------------------------------------------------------------------------------------------
// MFMA attention for the same problem (head_dim % 4 == 0): one WAVE per (batch, head), no workgroup barriers.
//   S^T = K Q^T   (keys x queries, 32x32x2 MFMAs, K and scaled Q staged in wave-private LDS)
//   softmax over keys = over the registers of a lane (+ one lane ^ 32 exchange): in the D fragment a lane owns ONE query column
//   O^T = V^T P^T : P^T is consumed straight out of the S^T accumulators as the MFMA B operand -- the contraction index is
//                   visited in the D fragment's own row order j(r, lane>>5) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), and the
//                   A operand reads V[j][d] from LDS in that same order, so the probabilities never move.
// DB = number of 32-wide blocks of the (zero-padded) head dimension.
// ------------------------------------------------------------------------------------------------
template <int DB, int TB>
__global__ __launch_bounds__(256) void cdx_attention_mfma_kernel(const cdx_attn_args a) {
    constexpr int TP = 32 * TB;                          // padded token count
    constexpr int DHP = 32 * DB, LD = DHP + 1;
    extern __shared__ float att_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.B * a.n_heads) return;                 // waves are independent (no barriers below)
    const int b = pair / a.n_heads, h = pair - b * a.n_heads;
    const int dh = a.head_dim, dm = a.n_heads * dh, T = a.T;
    float* Ks = att_lds + (size_t)wave * (2 * TP * LD);
    float* QVs = Ks + TP * LD;
    const float* base = a.qkv + (size_t)b * T * (3 * dm) + h * dh;
    const int lr = lane & 31, lk = lane >> 5;

    // global -> registers -> LDS with every load of a matrix in flight at once (a load-use-per-iteration loop costs one
    // memory round trip per 64 floats and made this kernel latency bound); 8*DB lanes cover one padded token row.
    constexpr int NV = TP * DHP / 4 / 64;                // float4 per lane per matrix
    auto load_mat = [&](float4 (&reg)[NV], int which, float mul) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = i * 64 + lane, tok = v / (DHP / 4), d = (v - tok * (DHP / 4)) * 4;
            const bool live = tok < T && d < dh;         // head_dim % 4 == 0: a float4 is all live or all padding
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) x = *reinterpret_cast<const float4*>(base + (size_t)tok * (3 * dm) + which * dm + d);
            reg[i] = make_float4(x.x * mul, x.y * mul, x.z * mul, x.w * mul);
        }
    };
    auto park = [&](float* dst, const float4 (&reg)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = i * 64 + lane, tok = v / (DHP / 4), d = (v - tok * (DHP / 4)) * 4;
            float* p = dst + tok * LD + d;
            p[0] = reg[i].x; p[1] = reg[i].y; p[2] = reg[i].z; p[3] = reg[i].w;
        }
    };
    float4 rq[NV], rk[NV], rv[NV];
    load_mat(rk, 1, 1.0f);
    load_mat(rq, 0, a.scale);
    load_mat(rv, 2, 1.0f);                               // V is needed last: its latency hides under the S^T MFMAs
    park(Ks, rk);
    park(QVs, rq);
    f32x16 s[TB][TB];                                    // [query block][key block]
#pragma unroll
    for (int q = 0; q < TB; ++q)
#pragma unroll
        for (int k = 0; k < TB; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[q][k][r] = 0.f;
#pragma unroll 4
    for (int st = 0; st < DHP / 2; ++st) {
        const int d = 2 * st + lk;
        float kk[TB], qq[TB];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            kk[i] = Ks[(32 * i + lr) * LD + d];
            qq[i] = QVs[(32 * i + lr) * LD + d];
        }
#pragma unroll
        for (int q = 0; q < TB; ++q)
#pragma unroll
            for (int k = 0; k < TB; ++k) s[q][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[k], qq[q], s[q][k], 0, 0, 0);
    }
    // V replaces Q in LDS (same wave: LDS operations retire in order, the reads above are done before these writes land)
    park(QVs, rv);
    float inv_den[TB];
#pragma unroll
    for (int q = 0; q < TB; ++q) {
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < TB; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = k * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                float sc = s[q][k][r];
                if (a.mask != nullptr && j < T && q * 32 + lr < T) sc += a.mask[(q * 32 + lr) * T + j];
                s[q][k][r] = j < T ? fmaxf(sc, -3.0e38f) : -3.0e38f;     // -inf entries become the finite floor: exp -> 0
                mx = fmaxf(mx, s[q][k][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < TB; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = k * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float e = j < T ? expf(s[q][k][r] - mx) : 0.f;
                s[q][k][r] = e;
                den += e;
            }
        den += __shfl_xor(den, 32, 64);
        inv_den[q] = 1.0f / den;
    }
    f32x16 o[TB][DB];
#pragma unroll
    for (int q = 0; q < TB; ++q)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[q][db][r] = 0.f;
#pragma unroll
    for (int k = 0; k < TB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = k * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const float v = QVs[j * LD + db * 32 + lr];
#pragma unroll
                for (int q = 0; q < TB; ++q) o[q][db] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[q][k][r], o[q][db], 0, 0, 0);
            }
        }
    // O^T fragment: column = query lr of block q, rows d = db*32 + (r & 3) + 8 (r >> 2) + 4 lk -> four float4 per block
#pragma unroll
    for (int q = 0; q < TB; ++q) {
        const int tok = q * 32 + lr;
        if (tok >= T) continue;
        float* op = a.out + ((size_t)b * T + tok) * dm + h * dh;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = db * 32 + 8 * g4 + 4 * lk;
                if (d0 < dh)
                    *reinterpret_cast<float4*>(op + d0) = make_float4(o[q][db][4 * g4] * inv_den[q], o[q][db][4 * g4 + 1] * inv_den[q],
                                                                       o[q][db][4 * g4 + 2] * inv_den[q], o[q][db][4 * g4 + 3] * inv_den[q]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Cross-attention against a short memory (S = 1 + n_obs <= 16 keys): one thread per (batch, head, query).  The work is
// tiny (T x S x head_dim per head); K/V rows are shared by the T threads of a (batch, head) and come from cache.
// ------------------------------------------------------------------------------------------------
#define XA_MAX_S 16
__global__ __launch_bounds__(256) void cdx_cross_attention_kernel(const cdx_xattn_args a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = a.B * a.n_heads * a.T;
    if (idx >= total) return;
    const int t = idx % a.T, h = (idx / a.T) % a.n_heads, b = idx / (a.T * a.n_heads);
    const int dh = a.head_dim, dm = a.n_heads * dh, S = 1 + a.n_obs;
    const float* q = a.q + ((size_t)b * a.T + t) * dm + h * dh;
    const float* kv[XA_MAX_S];
    kv[0] = a.kv_shared + (size_t)(a.shared_per_sample ? b : a.shared_row) * (2 * dm) + h * dh;
#pragma unroll
    for (int s = 1; s < XA_MAX_S; ++s)
        kv[s] = s < S ? a.kv_rows + ((size_t)b * a.n_obs + (s - 1)) * (2 * dm) + h * dh : kv[0];
    float sc[XA_MAX_S];
    float mx = -3.0e38f;
#pragma unroll
    for (int s = 0; s < XA_MAX_S; ++s) {
        float acc = 0.f;
        if (s < S) {
            for (int d = 0; d < dh; ++d) acc = fmaf(q[d], kv[s][d], acc);
            acc *= a.scale;
            if (a.mask) acc += a.mask[t * S + s];
            acc = fmaxf(acc, -3.0e38f);
        } else {
            acc = -3.0e38f;
        }
        sc[s] = acc;
        mx = fmaxf(mx, acc);
    }
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < XA_MAX_S; ++s) {
        sc[s] = s < S ? expf(sc[s] - mx) : 0.f;
        den += sc[s];
    }
    const float inv = 1.0f / den;
    float* o = a.out + ((size_t)b * a.T + t) * dm + h * dh;
    for (int d = 0; d < dh; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < XA_MAX_S; ++s)
            if (s < S) acc = fmaf(sc[s], kv[s][dm + d], acc);
        o[d] = acc * inv;
    }
}

// Coalesced variant (head_dim = 4 * GL, GL a power of two): GL adjacent lanes own one (row, head), each lane one float4 of the
// head -- q and out move as contiguous 16-byte lane accesses (the scalar kernel above reads a 1-KiB-strided row per thread
// and ran at ~0.3 TB/s); scores are reduced across the GL lanes with xor shuffles, the T rows of a sample hit the same K/V lines.
template <int GL>
__global__ __launch_bounds__(256) void cdx_cross_attention_vec_kernel(const cdx_xattn_args a) {
    const int dh = 4 * GL, dm = a.n_heads * dh, S = 1 + a.n_obs, cpr = dm >> 2;          // float4 chunks per row
    const long long rows = (long long)a.B * a.T;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    long long r = idx / cpr;
    const int c = (int)(idx - r * cpr);
    const bool live = r < rows;                     // a (row, head) group is live or dead as a whole: shuffles stay uniform
    if (!live) r = rows - 1;
    const int b = (int)(r / a.T), t = (int)(r - (long long)b * a.T);
    const int d0 = c * 4;
    const float4 q = *reinterpret_cast<const float4*>(a.q + (size_t)r * dm + d0);
    const float* kvs = a.kv_shared + (size_t)(a.shared_per_sample ? b : a.shared_row) * (2 * dm) + d0;
    const float* kvr = a.n_obs > 0 ? a.kv_rows + (size_t)b * a.n_obs * (2 * dm) + d0 : kvs;
    float sc[XA_MAX_S];
    float mx = -3.0e38f;
#pragma unroll
    for (int s = 0; s < XA_MAX_S; ++s) {
        float p = -3.0e38f;
        if (s < S) {
            const float4 k = *reinterpret_cast<const float4*>(s == 0 ? kvs : kvr + (size_t)(s - 1) * (2 * dm));
            p = fmaf(q.x, k.x, fmaf(q.y, k.y, fmaf(q.z, k.z, q.w * k.w)));
#pragma unroll
            for (int o = GL / 2; o >= 1; o >>= 1) p += __shfl_xor(p, o, 64);
            p *= a.scale;
            if (a.mask) p += a.mask[t * S + s];
            p = fmaxf(p, -3.0e38f);
        }
        sc[s] = p;
        mx = fmaxf(mx, p);
    }
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < XA_MAX_S; ++s) {
        sc[s] = s < S ? expf(sc[s] - mx) : 0.f;
        den += sc[s];
    }
    const float inv = 1.0f / den;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < XA_MAX_S; ++s)
        if (s < S) {
            const float4 v = *reinterpret_cast<const float4*>((s == 0 ? kvs : kvr + (size_t)(s - 1) * (2 * dm)) + dm);
            acc.x = fmaf(sc[s], v.x, acc.x); acc.y = fmaf(sc[s], v.y, acc.y);
            acc.z = fmaf(sc[s], v.z, acc.z); acc.w = fmaf(sc[s], v.w, acc.w);
        }
    if (live)
        *reinterpret_cast<float4*>(a.out + (size_t)r * dm + d0) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// ------------------------------------------------------------------------------------------------
// Elementwise unary map (SiLU / Mish / ...) for the batch-invariant embedding vectors
// ------------------------------------------------------------------------------------------------
__global__ void cdx_act_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = gm_act(x[i], act);
}

// d act(x) / dx, the factor of an explicit backward pass through an MLP (classifier guidance without autograd for the MLP / QGPO
// energy classifiers, reference classifier/base.py:74-79 over nn_classifier/mlp.py:10-55).  `param` is the scale s of a squashed
// output s * tanh(x / s) (QGPO: 10 tanh(out / 10)); 1 for plain tanh.
__device__ __forceinline__ float gm_act_grad(float x, int act, float param) {
    switch (act) {
        case CDX_ACT_NONE: return 1.0f;
        case CDX_ACT_RELU: return x > 0.f ? 1.0f : 0.f;
        case CDX_ACT_LEAKY: return x > 0.f ? 1.0f : 0.01f;
        case CDX_ACT_SILU: {                             // s (1 + x (1 - s)), s = sigmoid(x)
            const float sg = 1.0f / (1.0f + __expf(-x));
            return sg * (1.0f + x * (1.0f - sg));
        }
        case CDX_ACT_MISH: return gm_act(x, CDX_ACT_MISH_GRAD);
        case CDX_ACT_GELU_TANH: {                        // y = x s, s = sigmoid(2u), u = c (x + 0.044715 x^3):  s + x s (1 - s) 2c (1 + 0.134145 x^2)
            const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);
            const float sg = 1.0f / (1.0f + __expf(-u2));
            return sg + x * sg * (1.0f - sg) * 1.5957691216057308f * (1.0f + 0.134145f * x * x);
        }
        case CDX_ACT_TANH: {
            const float t = gm_act(x / param, CDX_ACT_TANH);
            return 1.0f - t * t;
        }
        case CDX_ACT_GELU_ERF: {                         // Phi(x) + x phi(x)
            const float z = fabsf(x) * 0.70710678118654752f;
            const float t = 1.0f / fmaf(0.3275911f, z, 1.0f);
            const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            const float erf_abs = 1.0f - poly * __expf(-z * z);
            return 0.5f * (1.0f + copysignf(erf_abs, x)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
        }
        default: return 1.0f;
    }
}
// out = g * act'(pre)
__global__ void cdx_act_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ g, float* __restrict__ out, size_t n, int act,
                                   float param) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = g[i] * gm_act_grad(pre[i], act, param);
}

static int gm_launch(const cdx_gemm_args* g, void* hip_stream, bool force_small, int* defer_slices = nullptr);
static bool gn_vec_enabled() {                       // CDX_GN_VEC=0: the scalar GroupNorm kernel everywhere (A/B hook)
    static const bool on = [] { const char* e = getenv("CDX_GN_VEC"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" {

int cdx_gemm_f32(const cdx_gemm_args* g, void* hip_stream) {
    if (!g) { cdx_set_err("cdx_gemm_f32: null argument block"); return CDX_EINVAL; }
    // N = 2.5 tiles (DiT / ChiTransformer projections with d_model 320): the 128-wide tiles would pad the last 64 columns to 128 (17 %
    // of the MFMAs wasted).  Large problems are cut in two launches instead: the multiple-of-128 part on 128 x 128 tiles, the
    // remainder (<= 64 columns) on the 64 x 64 kernel.  Measured on MI355X: M = 65536, N = 320: K = 1280 62.6 -> ~71 %, K = 320 49.5 -> ~57 %.
    static const char* env_sp = getenv("CDX_GEMM_SPLIT_N");
    if ((env_sp ? atoi(env_sp) : 1) && g->M >= 4096 && g->N > 128 && g->N % 128 != 0 && g->N % 128 <= 64 && g->N % 4 == 0 &&
        g->K % 32 == 0 && !g->table && g->conv_taps == 0 && g->A && g->W && g->C && g->K > 0) {
        const int n1 = g->N - g->N % 128;
        cdx_gemm_args a = *g, b = *g;
        a.N = n1;
        b.N = g->N - n1;
        b.W = g->W + (size_t)n1 * g->ldw;
        b.C = g->C + n1;
        if (g->bias) b.bias = g->bias + n1;
        if (g->gate) b.gate = g->gate + n1;
        if (g->residual) b.residual = g->residual + n1;
        a.partial = b.partial = nullptr;                 // (large M: split-K would not engage anyway)
        a.partial_slices = b.partial_slices = 0;
        // (Measured and dropped in round 5: the remainder on a side stream of the library NEXT TO the 128-wide launch -- registers and
        //  LDS would let one 4-wave workgroup sit beside the two 8-wave ones on a CU -- is 2 % SLOWER on the config-4 shard than the
        //  two launches back to back, profiles/r05_ln_vec_and_side_stream_ab.txt.)
        const int rc = gm_launch(&a, hip_stream, false);
        return rc != CDX_OK ? rc : gm_launch(&b, hip_stream, true);
    }
    return gm_launch(g, hip_stream, false);
}

}  // extern "C"

static int gm_launch(const cdx_gemm_args* g_in, void* hip_stream, bool force_small, int* defer_slices) {
    const cdx_gemm_args* g = g_in;
    if (!g) { cdx_set_err("cdx_gemm_f32: null argument block"); return CDX_EINVAL; }
    if (g->M < 0 || g->N <= 0 || g->K <= 0) { cdx_set_err("cdx_gemm_f32: bad shape"); return CDX_EINVAL; }
    if (g->M == 0) return CDX_OK;                                   // empty batch: nothing to launch
    if (!g->A || !g->W || !g->C) { cdx_set_err("cdx_gemm_f32: null pointer"); return CDX_EINVAL; }
    if ((g->gate && g->rows_per_gate <= 0) || (g->table && g->table_rows <= 0)) {
        cdx_set_err("cdx_gemm_f32: gate/table need a positive row period"); return CDX_EINVAL;
    }
    // tile shape: 128 x 128 (K tile 16) unless that leaves most of the chip idle -- then 64 x 64 (K tile 32)
    const int tiles_big = ((g->M + 127) / 128) * ((g->N + 127) / 128);
    static const char* env_t = getenv("CDX_GEMM_SMALL_TILE_BELOW");          // tuning hook
    // (with the row-run staging map the 64 x 64 tiles pay up to ~500 big tiles when the launch cannot split K instead -- ChiTransformer
    //  +3.5 %, DiT shards +0.5-2 %, profiles/r04_gemm_tile_threshold.txt; launches that CAN split K keep the 128 x 128 tiles: config 3 -8 %)
    // (round 6, on the 8-wave kernel: outputs wider than 64 columns take the 128 x 128 tiles from 256 tiles on -- ChiTransformer's N = 256
    //  projections at B = 1024: +4 %; narrow outputs (a 29-column head) keep the 64 x 64 tiles up to 520, profiles/r06_gemm_thresholds_ab.txt)
    const int small_below = env_t ? atoi(env_t) : (g->partial != nullptr && g->partial_slices > 1 ? 192 : (g->N <= 64 ? 520 : 256));
    // the 64 x 64 variant stages 32-wide K tiles: with K % 32 != 0 but K % 16 == 0 the 128 x 128 kernel keeps its unguarded loads
    const bool small = force_small || (tiles_big < small_below && (g->K % 32 == 0 || g->K % 16 != 0));
    // 128 x 128 tiles with unguarded loads take the 8-wave shape (K-blocked sums; four waves per SIMD); CDX_GEMM_W8=0: the 4-wave
    // shape everywhere (A/B hook).  Decided HERE because the K tile (hence the split-K arithmetic below) follows the shape.
    static const char* env_w8 = getenv("CDX_GEMM_W8");
    const bool aligned16 = (g->lda % 4 == 0) && (g->ldw % 4 == 0) && (((uintptr_t)g->A | (uintptr_t)g->W) % 16 == 0) &&
                           (g->conv_taps == 0 || g->conv_cin % 4 == 0);
    const bool w8 = CDX_GEMM_KBLOCK && !small && aligned16 && g->K % CDX_GEMM_W8_BK == 0 && g->K % 16 == 0 &&
                    (env_w8 ? atoi(env_w8) != 0 : CDX_GEMM_W8_DEFAULT);
    const int bmn = small ? 64 : 128, bk = small ? 32 : (w8 ? CDX_GEMM_W8_BK : 16);
    const int tiles = ((g->M + bmn - 1) / bmn) * ((g->N + bmn - 1) / bmn);
    if (g->conv_taps < 0 || (g->conv_taps > 0 && (g->conv_cin <= 0 || g->conv_lin <= 0 || g->conv_lout <= 0 || g->conv_stride <= 0 ||
                                                   g->K != g->conv_taps * g->conv_cin || g->M % g->conv_lout != 0))) {
        cdx_set_err("cdx_gemm_f32: inconsistent implicit-conv description"); return CDX_EINVAL;
    }
    const bool vec = (g->K % bk == 0) && aligned16;
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    // split-K when the tile count cannot fill the chip and K is long enough to pay for the second pass
    int k_split = 1;
    // resident workgroups of the chosen shape: 4 x 256 threads per CU (64 x 64 tiles), 3 (4-wave 128 x 128), 2 x 512 threads (8-wave 128 x 128:
    // the round-5 shape kept the 4-wave shape's 768 until round 6 -- a 256-tile conv of config 3 was cut into 3 slices = 1.5 rounds of workgroups)
    static const char* env_sl = getenv("CDX_GEMM_SPLITK_SLOTS");  // tuning hook
    const int slots = env_sl && atoi(env_sl) > 0 ? atoi(env_sl) : (small ? 1024 : (w8 ? 512 : 768));
    static const char* env_f = getenv("CDX_GEMM_SPLITK_FILL");     // tuning hook: split K while tiles fill less than this % of the slots
    const int fill = env_f ? atoi(env_f) : 51;      // (<= half: the GroupNorm-folded reduction made the second pass free; config 3 +1.4 %)
    if (g->partial != nullptr && g->partial_slices > 1 && (long long)tiles * 100 < (long long)slots * fill) {
        const int nk_all = (g->K + bk - 1) / bk;
        static const char* env_rd = getenv("CDX_GEMM_SPLITK_ROUND");   // tuning hook: "floor" = never more workgroups than slots
        k_split = (env_rd && env_rd[0] == 'f') ? (slots / tiles > 1 ? slots / tiles : 2) : (slots + tiles - 1) / tiles;
        if (k_split > g->partial_slices) k_split = g->partial_slices;
        static const char* env_mt = getenv("CDX_GEMM_SPLITK_MIN_TILES");     // tuning hook
        const int min_tiles = env_mt && atoi(env_mt) > 0 ? atoi(env_mt) : 8;
        if (k_split > nk_all / min_tiles) k_split = nk_all / min_tiles;      // at least 8 K tiles per slice
        if (k_split < 1) k_split = 1;
        const int per = (nk_all + k_split - 1) / k_split;
        k_split = (nk_all + per - 1) / per;                          // no empty slices
    }
    // deferred reduction (cdx_gemm_partials_f32): the raw K-slice sums stay in `partial` for the consumer to add up; an unsplit launch
    // writes its raw sums as slice 0
    cdx_gemm_args raw;
    if (defer_slices != nullptr) {
        *defer_slices = k_split;
        if (k_split == 1) {
            raw = *g;
            raw.C = g->partial; raw.ldc = g->N; raw.bias = nullptr; raw.gate = nullptr; raw.residual = nullptr; raw.table = nullptr;
            raw.act = CDX_ACT_NONE;
            g = &raw;
        }
    }
    const uintptr_t ep_ptrs = (uintptr_t)g->C | (uintptr_t)g->gate | (uintptr_t)g->residual | (uintptr_t)g->table;
    const int fast_ep = (g->N % 4 == 0) && (g->ldc % 4 == 0) && (!g->gate || g->ldg % 4 == 0) &&
                        (!g->residual || g->ldr % 4 == 0) && (ep_ptrs % 16 == 0);
    static const char* env_x = getenv("CDX_GEMM_XCD_ORDER");      // tuning hook: 1 = one contiguous tile range per XCD
    const int xcd_order = env_x ? atoi(env_x) : 1;    // rounds 3-5: +-2 % either way; on the round-6 build +1-2 % for DiT / ChiTransformer, neutral elsewhere
#if CDX_GEMM_PERSIST
    static const char* env_ps = getenv("CDX_GEMM_PERSIST_SLOTS");    // workgroups of a persistent launch (default: two per CU)
    const int slots_ps = env_ps && atoi(env_ps) > 0 ? atoi(env_ps) : 512;
    const dim3 grid(w8 && vec ? (tiles * k_split < slots_ps ? tiles * k_split : slots_ps) : tiles * k_split), block(w8 ? 512 : GM_THREADS);
#else
    const dim3 grid(tiles * k_split), block(w8 ? 512 : GM_THREADS);
#endif
    const size_t lds8 = CDX_GEMM_W8_BK == 32 ? (size_t)4 * 32 * 130 * sizeof(float) : 0;
    if (w8 && lds8) {
        static bool raised[2] = {false, false};
        const int ci = g->conv_taps > 0 ? 1 : 0;
        if (!raised[ci]) {
            const void* fn = ci ? reinterpret_cast<const void*>(cdx_gemm_kernel<true, 2, true, 8>) : reinterpret_cast<const void*>(cdx_gemm_kernel<true, 2, false, 8>);
            raised[ci] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8) == hipSuccess;
        }
    }
#define GM_LAUNCH(F, W, C) hipLaunchKernelGGL((cdx_gemm_kernel<F, W, C>), grid, block, 0, s, *g, fast_ep, k_split, xcd_order)
#if CDX_GEMM_KBLOCK
#define GM_LAUNCH8(C) hipLaunchKernelGGL((cdx_gemm_kernel<true, 2, C, 8>), grid, block, lds8, s, *g, fast_ep, k_split, xcd_order)
#else
#define GM_LAUNCH8(C) GM_LAUNCH(true, 2, C)
#endif
    const bool cv = g->conv_taps > 0;
    if (small) {
        if (vec) { if (cv) GM_LAUNCH(true, 1, true); else GM_LAUNCH(true, 1, false); }
        else { if (cv) GM_LAUNCH(false, 1, true); else GM_LAUNCH(false, 1, false); }
    } else if (w8) {
        if (cv) GM_LAUNCH8(true); else GM_LAUNCH8(false);
    } else {
        if (vec) { if (cv) GM_LAUNCH(true, 2, true); else GM_LAUNCH(true, 2, false); }
        else { if (cv) GM_LAUNCH(false, 2, true); else GM_LAUNCH(false, 2, false); }
    }
#undef GM_LAUNCH
#undef GM_LAUNCH8
    if (k_split > 1 && defer_slices == nullptr) {
        const size_t total = (size_t)g->M * g->N;
        const bool v4 = fast_ep && ((uintptr_t)g->partial % 16 == 0);
        const size_t work = v4 ? total / 4 : total;
        const int blocks = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
        if (v4) hipLaunchKernelGGL(gm_splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, *g, k_split);
        else hipLaunchKernelGGL(gm_splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, *g, k_split);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

// ---- library-internal entries (declared where they are used: csrc/cdx_bigbatch.hip) ------------------------------------------
// The conv GEMM of `g` with its K-slice reduction DEFERRED: the raw slice sums (no bias, no epilogue) are left in g->partial
// [slice][M][N], *slices says how many (>= 1; the split is chosen as for cdx_gemm_f32, up to g->partial_slices).  The consumer adds
// them up: cdx_groupnorm_slices_f32.
int cdx_gemm_partials_f32(const cdx_gemm_args* g, void* hip_stream, int* slices) {
    if (!g || !slices || !g->partial || g->partial_slices < 1) { cdx_set_err("cdx_gemm_partials_f32: bad argument"); return CDX_EINVAL; }
    if (g->gate || g->residual || g->table || g->act != CDX_ACT_NONE || g->N % 4 != 0) {
        cdx_set_err("cdx_gemm_partials_f32: epilogue terms cannot be deferred"); return CDX_EINVAL;
    }
    cdx_gemm_args q = *g;
    q.C = g->partial;                 // (never written in a split launch; an unsplit one writes slice 0 through it)
    q.ldc = g->N;
    return gm_launch(&q, hip_stream, false, slices);
}

// Shapes the slice-summing GroupNorm takes (the float4 kernel with the group in registers).
bool cdx_groupnorm_slices_ok(int L, int C, int G) {
    if (G <= 0 || C % G != 0) return false;
    const int cg = C / G;
    return gn_vec_enabled() && cg % 4 == 0 && (long long)L * cg <= 64LL * 4 * GN_VREGS && C % 4 == 0;
}

// cdx_groupnorm_f32 on x = sum over `slices` partial tensors (a->x, a->x + slice_stride, ...) + xbias[channel].
int cdx_groupnorm_slices_f32(const cdx_gn_args* a, int slices, long long slice_stride, const float* xbias, void* hip_stream) {
    if (!a || slices < 1 || slice_stride < 0) { cdx_set_err("cdx_groupnorm_slices_f32: bad argument"); return CDX_EINVAL; }
    if (a->B == 0) return CDX_OK;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!a->x || !a->y || !a->gamma || !a->beta || a->B < 0 || a->L <= 0 || !cdx_groupnorm_slices_ok(a->L, a->C, a->G) || a->ldx % 4 != 0 ||
        a->ldy % 4 != 0 || slice_stride % 4 != 0 || !al16(a->x) || !al16(a->y) || !al16(a->gamma) || !al16(a->beta) || !al16(xbias) ||
        (a->residual && (a->ldr % 4 != 0 || !al16(a->residual))) || (a->fa && (a->ldfa % 4 != 0 || !al16(a->fa))) ||
        (a->fb && (a->ldfb % 4 != 0 || !al16(a->fb))) || a->film_mode < 0 || a->film_mode > 2) {
        cdx_set_err("cdx_groupnorm_slices_f32: shape or alignment outside the float4 kernel"); return CDX_EINVAL;
    }
    const long long waves = (long long)a->B * a->G;
    hipLaunchKernelGGL(cdx_groupnorm_vec_kernel<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(hip_stream), *a, slices, (size_t)slice_stride, xbias);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

extern "C" {

int cdx_gemm_set_trace(unsigned long long* device_buffer) {
    unsigned long long* p = device_buffer;
    if (hipMemcpyToSymbol(HIP_SYMBOL(gm_trace), &p, sizeof(p)) != hipSuccess) { cdx_set_err("cdx_gemm_set_trace failed"); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_layernorm_f32(const cdx_ln_args* a, void* hip_stream) {
    if (!a) { cdx_set_err("cdx_layernorm_f32: null argument block"); return CDX_EINVAL; }
    if (a->M > 0 && (!a->x || !a->y)) { cdx_set_err("cdx_layernorm_f32: null pointer"); return CDX_EINVAL; }
    if (a->C <= 0 || a->C > 4096 || a->M < 0) { cdx_set_err("cdx_layernorm_f32: 0 < C <= 4096 required"); return CDX_EINVAL; }
    if ((a->scale != nullptr) != (a->shift != nullptr) || (a->gamma != nullptr) != (a->beta != nullptr)) {
        cdx_set_err("cdx_layernorm_f32: scale/shift and gamma/beta come in pairs"); return CDX_EINVAL;
    }
    if (a->scale && a->rows_per_mod <= 0) { cdx_set_err("cdx_layernorm_f32: rows_per_mod must be positive"); return CDX_EINVAL; }
    if (a->M == 0) return CDX_OK;
    cdx_ln_args b = *a;
    if (b.rows_per_mod <= 0) b.rows_per_mod = 1;
    // rows of up to 1024 columns with 16-byte aligned everything: the float4 kernel (CDX_LN_VEC=0: the scalar one everywhere, A/B hook)
    static const bool vec_on = [] { const char* e = getenv("CDX_LN_VEC"); return !(e && e[0] == '0'); }();
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec = vec_on && a->C <= 1024 && a->C % 4 == 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0 && al16(a->x) && al16(a->y) &&
                     al16(a->gamma) && al16(a->beta) && al16(a->scale) && al16(a->shift) && (!a->scale || a->ldmod % 4 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    if (vec) {
        const dim3 grid((a->M + 15) / 16), block(256);
        if (a->C <= 256) hipLaunchKernelGGL(cdx_layernorm_vec_kernel<4>, grid, block, 0, st, b);
        else if (a->C <= 512) hipLaunchKernelGGL(cdx_layernorm_vec_kernel<8>, grid, block, 0, st, b);
        else hipLaunchKernelGGL(cdx_layernorm_vec_kernel<16>, grid, block, 0, st, b);
    } else {
        auto kern = a->C <= 1024 ? cdx_layernorm_kernel<16> : (a->C <= 2048 ? cdx_layernorm_kernel<32> : cdx_layernorm_kernel<64>);
        hipLaunchKernelGGL(kern, dim3((a->M + 3) / 4), dim3(256), 0, st, b);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_groupnorm_f32(const cdx_gn_args* a, void* hip_stream) {
    if (!a) { cdx_set_err("cdx_groupnorm_f32: null argument block"); return CDX_EINVAL; }
    if (a->B < 0 || a->L <= 0 || a->C <= 0 || a->G <= 0 || a->C % a->G != 0 || a->film_mode < 0 || a->film_mode > 2) {
        cdx_set_err("cdx_groupnorm_f32: bad shape (C must be a multiple of G)"); return CDX_EINVAL;
    }
    if (a->B == 0) return CDX_OK;
    if (!a->x || !a->y || !a->gamma || !a->beta) { cdx_set_err("cdx_groupnorm_f32: null pointer"); return CDX_EINVAL; }
    const long long waves = (long long)a->B * a->G;
    const int cg = a->C / a->G;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = gn_vec_enabled() && cg % 4 == 0 && (long long)a->L * cg <= 64LL * 4 * GN_VREGS && a->ldx % 4 == 0 && a->ldy % 4 == 0 && a->C % 4 == 0 &&
                     al16(a->x) && al16(a->y) && al16(a->gamma) && al16(a->beta) &&
                     (!a->residual || (a->ldr % 4 == 0 && al16(a->residual))) && (!a->fa || (a->ldfa % 4 == 0 && al16(a->fa))) &&
                     (!a->fb || (a->ldfb % 4 == 0 && al16(a->fb)));
    if (vec) hipLaunchKernelGGL(cdx_groupnorm_vec_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), *a, 1, (size_t)0, nullptr);
    else hipLaunchKernelGGL(cdx_groupnorm_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), *a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_groupnorm_bwd_f32(const cdx_gn_args* a, void* hip_stream) {
    if (!a) { cdx_set_err("cdx_groupnorm_bwd_f32: null argument block"); return CDX_EINVAL; }
    if (a->B < 0 || a->L <= 0 || a->C <= 0 || a->G <= 0 || a->C % a->G != 0 || (a->act != CDX_ACT_MISH && a->act != CDX_ACT_NONE)) {
        cdx_set_err("cdx_groupnorm_bwd_f32: bad shape or unsupported activation"); return CDX_EINVAL;
    }
    if (a->B == 0) return CDX_OK;
    if (!a->x || !a->y || !a->gamma || !a->beta || !a->residual) { cdx_set_err("cdx_groupnorm_bwd_f32: null pointer"); return CDX_EINVAL; }
    if ((a->dgamma_part == nullptr) != (a->dbeta_part == nullptr)) { cdx_set_err("cdx_groupnorm_bwd_f32: dgamma_part and dbeta_part go together"); return CDX_EINVAL; }
    if ((a->dgamma_sum == nullptr) != (a->dbeta_sum == nullptr) || (a->dgamma_sum != nullptr && a->dgamma_part != nullptr)) {
        cdx_set_err("cdx_groupnorm_bwd_f32: dgamma_sum and dbeta_sum go together, and not with the *_part pair"); return CDX_EINVAL;
    }
    const int cg = a->C / a->G;
    const bool pow2 = cg <= 256 && (cg & (cg - 1)) == 0;
    if ((a->dgamma_part != nullptr || a->dgamma_sum != nullptr || a->dy_possum != nullptr) && !pow2) {
        cdx_set_err("cdx_groupnorm_bwd_f32: parameter gradients / position sums need a power-of-two group width <= 256"); return CDX_EINVAL;
    }
    if (a->dy_possum != nullptr && a->ld_possum < a->C) { cdx_set_err("cdx_groupnorm_bwd_f32: ld_possum < C"); return CDX_EINVAL; }
    const long long waves = (long long)a->B * a->G;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // the group-in-registers kernel: power-of-two groups of 4..256 channels, at most 8 float4 per lane, 16-byte rows (CDX_GN_VEC=0: off)
    const bool vec = gn_vec_enabled() && pow2 && cg >= 4 && (long long)a->L * cg <= 64LL * 4 * GN_VREGS && a->ldx % 4 == 0 && a->ldy % 4 == 0 &&
                     a->ldr % 4 == 0 && al16(a->x) && al16(a->y) && al16(a->residual) && al16(a->gamma) && al16(a->beta) &&
                     (a->dgamma_part == nullptr || (al16(a->dgamma_part) && al16(a->dbeta_part) && a->C % 4 == 0)) &&
                     (a->dy_possum == nullptr || (al16(a->dy_possum) && a->ld_possum % 4 == 0));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const dim3 grid_s((unsigned)(((a->B + 3) / 4) * (long long)a->G)), grid_p((unsigned)((waves + 3) / 4));
    if (a->dgamma_sum != nullptr) {
        // samples per wave of the register kernel: 1 -- measured on config 2 / 3 (profiles/r06_gn_bwd_vec_ab.txt): 2 / 4 / 8 samples per wave
        // cut the float atomics per address accordingly and are 10 / 60 / 170 % SLOWER, the launch is bound by one sample's dependent
        // chain (load -> two reductions -> Mish' -> reduction -> store), not by the atomics.  CDX_GN_BWD_SPW: the A/B hook
        static const char* env_spw = getenv("CDX_GN_BWD_SPW");
        int spw = env_spw ? atoi(env_spw) : 1;
        if (spw < 1 || spw > 64) spw = 1;
        const dim3 grid_v((unsigned)(((a->B + 4 * spw - 1) / (4 * spw)) * (long long)a->G));
        if (vec) hipLaunchKernelGGL(cdx_groupnorm_bwd_vec_kernel<true>, grid_v, dim3(256), 0, st, *a, spw);
        else hipLaunchKernelGGL(cdx_groupnorm_bwd_kernel<true>, grid_s, dim3(256), 0, st, *a);
    } else {
        if (vec) hipLaunchKernelGGL(cdx_groupnorm_bwd_vec_kernel<false>, grid_p, dim3(256), 0, st, *a, 1);
        else hipLaunchKernelGGL(cdx_groupnorm_bwd_kernel<false>, grid_p, dim3(256), 0, st, *a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_attention_f32(const cdx_attn_args* a, void* hip_stream) {
    if (!a) { cdx_set_err("cdx_attention_f32: null argument block"); return CDX_EINVAL; }
    if (a->B > 0 && (!a->qkv || !a->out)) { cdx_set_err("cdx_attention_f32: null pointer"); return CDX_EINVAL; }
    if (a->T <= 0 || a->T > CDX_ATTN_MAX_T || a->head_dim <= 0 || a->head_dim > 64 || a->n_heads <= 0 || a->B < 0) {
        cdx_set_err("cdx_attention_f32: T <= 1024 and head_dim <= 64 required"); return CDX_EINVAL;
    }
    if (a->B == 0) return CDX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int dm = a->n_heads * a->head_dim;
    if (a->T > 64) {                                     // streamed keys / online softmax
        const long long grid = (long long)a->B * a->n_heads * ((a->T + 63) / 64);
        if (grid > 0x7fffffffLL) { cdx_set_err("cdx_attention_f32: batch too large for one launch"); return CDX_EINVAL; }
        hipLaunchKernelGGL(cdx_attention_long_kernel, dim3((unsigned)grid), dim3(64), 0, st, *a);
    } else if (a->head_dim % 4 == 0 && dm % 4 == 0 && ((uintptr_t)a->out % 16) == 0) {      // MFMA path, one wave per (batch, head)
        const int pairs = a->B * a->n_heads, grid = (pairs + 3) / 4;
        // <head-dim blocks, token blocks>: T <= 32 runs the single-block variant (a quarter of the MFMAs, half the LDS)
        const int tb = a->T <= 32 ? 1 : 2, db = a->head_dim <= 32 ? 1 : 2;
        static bool lds_raised[2][2] = {{false, false}, {false, false}};
        auto launch = [&](auto kern, size_t lds) {
            if (lds > 48 * 1024 && !lds_raised[db - 1][tb - 1]) {
                lds_raised[db - 1][tb - 1] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, *a);
        };
        const size_t lds = (size_t)4 * 2 * (32 * tb) * (32 * db + 1) * sizeof(float);
        if (db == 1 && tb == 1) launch(cdx_attention_mfma_kernel<1, 1>, lds);
        else if (db == 1) launch(cdx_attention_mfma_kernel<1, 2>, lds);
        else if (tb == 1) launch(cdx_attention_mfma_kernel<2, 1>, lds);
        else launch(cdx_attention_mfma_kernel<2, 2>, lds);
    } else {
        hipLaunchKernelGGL(cdx_attention_kernel, dim3(a->B * a->n_heads), dim3(64), 0, st, *a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_cross_attention_f32(const cdx_xattn_args* a, void* hip_stream) {
    if (!a) { cdx_set_err("cdx_cross_attention_f32: null argument block"); return CDX_EINVAL; }
    if (a->B < 0 || a->T <= 0 || a->n_obs < 0 || 1 + a->n_obs > XA_MAX_S || a->n_heads <= 0 || a->head_dim <= 0) {
        cdx_set_err("cdx_cross_attention_f32: 1 + n_obs <= 16 memory tokens required"); return CDX_EINVAL;
    }
    if (a->B == 0) return CDX_OK;
    if (!a->q || !a->kv_shared || !a->out || (a->n_obs > 0 && !a->kv_rows)) { cdx_set_err("cdx_cross_attention_f32: null pointer"); return CDX_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int gl = a->head_dim / 4;
    const bool aligned = (((uintptr_t)a->q | (uintptr_t)a->out | (uintptr_t)a->kv_shared | (uintptr_t)a->kv_rows) % 16) == 0;
    if (aligned && a->head_dim % 4 == 0 && gl <= 64 && (gl & (gl - 1)) == 0) {
        const long long lanes = (long long)a->B * a->T * a->n_heads * gl;
        const dim3 grid((unsigned)((lanes + 255) / 256));
        switch (gl) {
            case 1: hipLaunchKernelGGL(cdx_cross_attention_vec_kernel<1>, grid, dim3(256), 0, st, *a); break;
            case 2: hipLaunchKernelGGL(cdx_cross_attention_vec_kernel<2>, grid, dim3(256), 0, st, *a); break;
            case 4: hipLaunchKernelGGL(cdx_cross_attention_vec_kernel<4>, grid, dim3(256), 0, st, *a); break;
            case 8: hipLaunchKernelGGL(cdx_cross_attention_vec_kernel<8>, grid, dim3(256), 0, st, *a); break;
            case 16: hipLaunchKernelGGL(cdx_cross_attention_vec_kernel<16>, grid, dim3(256), 0, st, *a); break;
            case 32: hipLaunchKernelGGL(cdx_cross_attention_vec_kernel<32>, grid, dim3(256), 0, st, *a); break;
            default: hipLaunchKernelGGL(cdx_cross_attention_vec_kernel<64>, grid, dim3(256), 0, st, *a); break;
        }
    } else {
        const long long total = (long long)a->B * a->n_heads * a->T;
        hipLaunchKernelGGL(cdx_cross_attention_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_act_bwd_f32(const float* pre, const float* g, float* out, long long n, int act, float param, void* hip_stream) {
    if (!pre || !g || !out || n < 0) { cdx_set_err("cdx_act_bwd_f32: bad argument"); return CDX_EINVAL; }
    if (act == CDX_ACT_MISH_GRAD || act < 0 || act > CDX_ACT_TANH || !(param != 0.f)) {
        cdx_set_err("cdx_act_bwd_f32: no derivative for this activation id / zero scale"); return CDX_EINVAL;
    }
    if (n == 0) return CDX_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(cdx_act_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), pre, g, out, (size_t)n,
                       act, param);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_act_f32(const float* x, float* y, long long n, int act, void* hip_stream) {
    if (!x || !y || n < 0) { cdx_set_err("cdx_act_f32: bad argument"); return CDX_EINVAL; }
    if (n == 0) return CDX_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(cdx_act_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), x, y,
                       (size_t)n, act);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

}  // extern "C"
