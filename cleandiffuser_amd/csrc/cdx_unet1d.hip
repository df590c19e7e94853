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
#define CDX_EPI_REGS 4  // GroupNorm elements a lane keeps in registers (groups of <= 256 elements)

static thread_local char g_err[256] = "";
void cdx_set_err(const char* msg) {          // shared with the other translation units of libcdx.so
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
static void set_err(const char* msg) { cdx_set_err(msg); }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)) with tanh(log(1+e^x)) = n / (n + 2), n = e^x (e^x + 2); softplus threshold 20 as ATen.
    // v_exp_f32 / v_rcp_f32 are 1-ulp instructions: relative error of the result ~3e-7, far inside the 1e-4 budget.
    const float e = __expf(fminf(x, 20.0f));
    const float n = e * (e + 2.0f);
    return x > 20.0f ? x : x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}

// Activation ids of csrc/cdx_ops.h (CDX_ACT_*).  `act` is wave-uniform, so the switch is a scalar branch.
template <bool FULL>
__device__ __forceinline__ float act_f(float x, int act) {
    if constexpr (!FULL) return act == CDX_ACT_MISH ? mish_f(x) : x;   // the U-Net programs only use Mish / identity
    switch (act) {
        case CDX_ACT_MISH: return mish_f(x);
        case CDX_ACT_GELU_ERF: {                         // erf by Abramowitz-Stegun 7.1.26, |err| < 1.5e-7, branch-free (as in cdx_gemm.hip)
            const float z = fabsf(x) * 0.70710678118654752f;
            const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
            const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            return 0.5f * x * (1.0f + copysignf(1.0f - poly * __expf(-z * z), x));
        }
        case CDX_ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
        case CDX_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
        case CDX_ACT_RELU: return fmaxf(x, 0.f);
        case CDX_ACT_GELU_TANH: return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
        default: return x;
    }
}

// wave64 all-reduce (sum) on the DPP network: quad swaps, row half-mirror, row mirror, then the 4 row sums
// are combined through SGPRs.  ~10 issue slots instead of 6 dependent ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_step(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_step<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_step<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_step<0x141>(v);   // row_half_mirror
    v = dpp_step<0x140>(v);   // row_mirror  -> every lane holds its 16-lane row sum
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Profiling stamps are written to LDS (a global store would be waited on by the next s_waitcnt vmcnt(0) and distort
// the phase being measured) and copied out once at kernel end.
__device__ __forceinline__ void stamp(unsigned long long* slot, int tid) {
    if (slot && tid == 0) *slot = __builtin_amdgcn_s_memtime();
}

// Input row feeding output position `pos` at kernel tap `tap`, or -1 when that tap contributes nothing (outside
// [0, l_in), odd phase of a stride-2 transposed conv, column past l_out).  Such lanes are pointed at the shared
// all-zero row, so the B fetch needs no predicate.  Transposed convs are stride 2 (host asserts).
__device__ __forceinline__ int conv_row(int pos, int tap, int cstride, int cpad, int transposed, int l_out, int l_in) {
    const int fwd = pos * cstride + tap - cpad;
    const int num = pos + cpad - tap;
    const int bwd = (num & 1) ? -1 : (num >> 1);
    const int q = transposed ? bwd : fwd;
    return (pos < l_out && q >= 0 && q < l_in) ? q : -1;
}

struct ConvGeom {
    int taps, cstride, cpad, transposed, l_out, l_in;
    int zrow;      // LDS offset of the shared all-zero row
    int col0;      // first output position of this pass (long horizons are covered in passes of 2 column tiles)
    int srcA, strideA, ca, srcB, strideB, cb;
};

struct ChunkCursor {           // scalar (SGPR) walk over K = (source, tap, channel chunk)
    int tap, cc, ccn, src, sstr;
};

// MFMA shape traits.  M16: v_mfma_f32_16x16x4_f32 -- 16 rows x 16 cols, a record covers 16 K values (4 per k-lane
// group).  M4: v_mfma_f32_4x4x1_16b_f32 -- 16 blocks of 4x4 = 64 rows x 4 cols, a record covers 4 K values.
struct M16 {
    static constexpr int COLS = 16, KSTEP = 16, PF = 8;
    static __device__ __forceinline__ int col(int lane) { return lane & 15; }
    static __device__ __forceinline__ int koff(int lane) { return 4 * (lane >> 4); }   // B: which 4 of the 16 K
    static __device__ __forceinline__ int drow(int lane) { return 4 * (lane >> 4); }   // D: first of 4 rows
    static __device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
struct M4 {
    static constexpr int COLS = 4, KSTEP = 4, PF = 8;
    static __device__ __forceinline__ int col(int lane) { return lane & 3; }
    static __device__ __forceinline__ int koff(int) { return 0; }
    static __device__ __forceinline__ int drow(int lane) { return 4 * (lane >> 2); }
    static __device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    }
};

template <class M, int NT>
__device__ __forceinline__ void lane_rows(const ConvGeom& g, int tap, int src, int sstr, int lane, int (&roff)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = conv_row(g.col0 + nt * M::COLS + M::col(lane), tap, g.cstride, g.cpad, g.transposed, g.l_out,
                               g.l_in);
        roff[nt] = (q >= 0 ? q * sstr : g.zrow - src) + M::koff(lane);     // offset relative to the source slot
    }
}

template <class M, int NT>
__device__ __forceinline__ void fetch_b(const float* __restrict__ lds, const ChunkCursor& c, const int (&roff)[NT],
                                        f32x4 (&bv)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        bv[nt] = *reinterpret_cast<const f32x4*>(lds + c.src + roff[nt] + c.cc * M::KSTEP);
}

// K loop of one conv: every wave walks its work items (item table built by the host: one row tile x one K range).
// All control flow is scalar (wave index via readfirstlane, item records via s_load), the weight ring keeps M::PF
// 1-KiB records in flight per wave with counted vmcnt, and chunk q+1's B operand is read from LDS under chunk q's
// MFMAs.
// Prefetched head of the next conv's weight stream: the first CDX_PRE records of this wave's first work item,
// issued before the previous op's barrier/epilogue so that they have landed when the K loop starts.
#define CDX_PRE 8
struct Prefetch {
    f32x4 rec[CDX_PRE];   // native vector type: SROA keeps these in VGPRs across the op loop
    int ok;               // wave-uniform: rec[] belongs to (this op, item == wave)
};

#define CDX_RL(v, k) __builtin_amdgcn_readlane((v), (k))

template <class M, int NT>
__device__ __forceinline__ void conv_kloop(const ConvGeom& g, const float* __restrict__ wblob,
                                           const int* __restrict__ items, int n_items,
                                           float* __restrict__ lds, int scratch, int sstride, int lane, int wave,
                                           Prefetch& pre, unsigned long long* prof) {
    constexpr int PF = M::PF;
    static_assert(PF >= CDX_PRE, "ring shallower than the cross-op prefetch");
    const int tid0 = (wave == 0 && lane == 0) ? 0 : 1;       // stamp() fires for tid == 0 only
    for (int item = wave; item < n_items; item += CDX_N_WAVES) {
        const int iw = items[item * CDX_ITEM_WORDS + (lane & 7)];   // item record: one (flat) load, then SGPRs
        const int it0 = CDX_RL(iw, CDX_I_WOFF), it1 = CDX_RL(iw, CDX_I_PART), nq = CDX_RL(iw, CDX_I_NQ);
        const int it3 = CDX_RL(iw, CDX_I_ONB), it4 = CDX_RL(iw, CDX_I_TAP), it5 = CDX_RL(iw, CDX_I_CC);
        if (prof && item == 0) { asm volatile("" ::"s"(nq)); stamp(prof + 4, tid0); }
        ChunkCursor c;
        c.tap = it4; c.cc = it5;
        if (it3) { c.ccn = g.cb; c.src = g.srcB; c.sstr = g.strideB; }
        else       { c.ccn = g.ca; c.src = g.srcA; c.sstr = g.strideA; }
        f32x4 acc0[NT], acc1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc0[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        int roff[NT];
        lane_rows<M, NT>(g, c.tap, c.src, c.sstr, lane, roff);

        const f32x4* wp = reinterpret_cast<const f32x4*>(wblob + it0) + lane;
        f32x4 wr[PF];
        if (pre.ok && item == wave) {                      // head of the stream was fetched during the previous op
#pragma unroll
            for (int u = 0; u < CDX_PRE; ++u) wr[u] = pre.rec[u];
#pragma unroll
            for (int u = CDX_PRE; u < PF; ++u) wr[u] = wp[(size_t)min(u, nq - 1) * 64];
        } else {
#pragma unroll
            for (int u = 0; u < PF; ++u) wr[u] = wp[(size_t)min(u, nq - 1) * 64];  // unconditional: vmcnt stays countable
        }
        f32x4 bcur[NT];
        fetch_b<M, NT>(lds, c, roff, bcur);
        if (prof && item == 0) { asm volatile("" ::"v"(wr[0][0]), "v"(bcur[0][0])); stamp(prof + 5, tid0); }

        auto chunk = [&](const f32x4 a, bool more) {
            if (++c.cc == c.ccn) {
                c.cc = 0;
                if (++c.tap == g.taps) {
                    c.tap = 0; c.ccn = g.cb; c.src = g.srcB; c.sstr = g.strideB;
                }
                lane_rows<M, NT>(g, c.tap, c.src, c.sstr, lane, roff);
            }
            f32x4 bnext[NT];
            if (more) fetch_b<M, NT>(lds, c, roff, bnext);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc0[nt] = M::mfma(a[0], bcur[nt][0], acc0[nt]);
                acc1[nt] = M::mfma(a[1], bcur[nt][1], acc1[nt]);
                acc0[nt] = M::mfma(a[2], bcur[nt][2], acc0[nt]);
                acc1[nt] = M::mfma(a[3], bcur[nt][3], acc1[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnext[nt];
        };

        // steady state: every ring slot is refilled unconditionally -> counted s_waitcnt vmcnt(PF-1)
        int qi = 0;
        const int n_main = (nq / PF - 1) * PF;
        for (; qi < n_main; qi += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                chunk(wr[u], true);
                wr[u] = wp[(size_t)(qi + u + PF) * 64];
            }
        }
        // drain: the last (up to 2*PF - 1) chunks, refilling only while records remain
        for (; qi < nq; qi += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (qi + u < nq) {
                    chunk(wr[u], qi + u + 1 < nq);
                    if (qi + u + PF < nq) wr[u] = wp[(size_t)(qi + u + PF) * 64];
                }
            }
        }
        if (prof && item == 0) { asm volatile("" ::"v"(acc0[0][0]), "v"(acc1[0][0])); stamp(prof + 6, tid0); }
        // D fragment: this lane holds 4 consecutive rows of one column -> scratch[k-slice][n][row tile + rows]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = g.col0 + nt * M::COLS + M::col(lane);
            if (n < g.l_out) {
                const f32x4 d = acc0[nt] + acc1[nt];
                *reinterpret_cast<float4*>(lds + scratch + it1 + n * sstride + M::drow(lane)) =
                    make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

// exact floor(e / d) for 0 <= e < 2^20 from a host-computed fp32 reciprocal (no integer division on the device)
__device__ __forceinline__ int div_small(int e, int d, float inv_d) {
    int q = (int)(((float)e + 0.5f) * inv_d);
    const int r = e - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

// `w` holds this op's descriptor, one word per lane (lane k = word k); `wn` the next op's.
// FULL = false compiles the lean U-Net instance (Mish only, no per-column norm / fill ops): the kernel is
// instruction-cache sensitive, so the MLP-tile features live in a second instantiation.
template <bool FULL>
__device__ __forceinline__ void conv_op(const int w, const int wn, const int* __restrict__ itab,
                        const float* __restrict__ wblob, float* __restrict__ lds,
                        int scratch, int zrow, int pred_branch_off, int tid, Prefetch& pre,
                        unsigned long long* prof) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_out = CDX_RL(w, CDX_W_COUT), c16 = CDX_RL(w, CDX_W_COUT16), l_out = CDX_RL(w, CDX_W_LOUT);
    const int flags = CDX_RL(w, CDX_W_FLAGS);
    const int dst = CDX_RL(w, CDX_W_DST) + ((flags & CDX_F_DST_PRED) ? pred_branch_off : 0);
    const int dstride = CDX_RL(w, CDX_W_DST_STRIDE), drows = CDX_RL(w, CDX_W_DST_ROWS);
    const int ksplit = CDX_RL(w, CDX_W_KSPLIT);
    const int sstride = c16 + 4;

    ConvGeom g;
    g.taps = CDX_RL(w, CDX_W_TAPS); g.cstride = CDX_RL(w, CDX_W_CSTRIDE); g.cpad = CDX_RL(w, CDX_W_CPAD);
    g.transposed = CDX_RL(w, CDX_W_TRANSPOSED); g.l_out = l_out; g.col0 = 0;
    g.l_in = CDX_RL(w, CDX_W_LIN); g.zrow = zrow;
    g.srcA = CDX_RL(w, CDX_W_SRCA); g.strideA = CDX_RL(w, CDX_W_SRCA_STRIDE); g.ca = CDX_RL(w, CDX_W_CA_CHUNKS);
    g.srcB = CDX_RL(w, CDX_W_SRCB); g.strideB = CDX_RL(w, CDX_W_SRCB_STRIDE); g.cb = CDX_RL(w, CDX_W_CB_CHUNKS);

    // 1. clear the destination slot (halo rows + pad columns must read as zero for the consumer)
    if (!(flags & (CDX_F_ACCUM | CDX_F_KEEP_DST))) {
        const int total = drows * dstride;  // multiple of 4, dst 16-byte aligned
        for (int i = tid * 4; i < total; i += CDX_THREADS * 4)
            *reinterpret_cast<float4*>(lds + dst + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // 2. implicit-GEMM K loop -> split-K partials in scratch
    const int* __restrict__ items = itab + CDX_RL(w, CDX_W_ITEMS);
    const int n_items = CDX_RL(w, CDX_W_NITEMS);
    if (CDX_RL(w, CDX_W_MODE) == CDX_MODE_4X4) {
        if (l_out <= 4) conv_kloop<M4, 1>(g, wblob, items, n_items, lds, scratch, sstride, lane, wave, pre, prof);
        else conv_kloop<M4, 2>(g, wblob, items, n_items, lds, scratch, sstride, lane, wave, pre, prof);
    } else {
        if (l_out <= 16) conv_kloop<M16, 1>(g, wblob, items, n_items, lds, scratch, sstride, lane, wave, pre, prof);
        else {
            // 32 positions per pass; longer horizons re-stream the weights once per extra pass (rare: H = 64)
            for (g.col0 = 0; g.col0 < l_out; g.col0 += 32) {
                conv_kloop<M16, 2>(g, wblob, items, n_items, lds, scratch, sstride, lane, wave, pre, prof);
                pre.ok = 0;
            }
        }
    }

    // 2b. cross-op prefetch: start streaming the head of the NEXT conv's weights for this wave's first item now;
    //     the loads fly through the barrier and the epilogue below.
    pre.ok = 0;
    if (CDX_RL(wn, CDX_W_KIND) == CDX_OP_CONV && wave < CDX_RL(wn, CDX_W_NITEMS)) {
        const int iw = itab[CDX_RL(wn, CDX_W_ITEMS) + wave * CDX_ITEM_WORDS + (lane & 7)];
        const int nq = CDX_RL(iw, CDX_I_NQ);
        const f32x4* wp = reinterpret_cast<const f32x4*>(wblob + CDX_RL(iw, CDX_I_WOFF)) + lane;
#pragma unroll
        for (int u = 0; u < CDX_PRE; ++u) pre.rec[u] = wp[(size_t)min(u, nq - 1) * 64];
        pre.ok = 1;
    }
    stamp(prof ? prof + 7 : nullptr, tid);
    // 3. epilogue.  Per-channel parameters are fetched BEFORE the barrier so their latency hides behind the wait
    //    for the slowest wave; everything after the barrier touches only LDS.
    const float* __restrict__ bias = wblob + CDX_RL(w, CDX_W_BOFF);
    const int res = CDX_RL(w, CDX_W_RES), rstride = CDX_RL(w, CDX_W_RES_STRIDE), emb = CDX_RL(w, CDX_W_EMB);
    const int groups = CDX_RL(w, CDX_W_GROUPS), cg = CDX_RL(w, CDX_W_CG), sh = CDX_RL(w, CDX_W_CG_SHIFT);
    const int cnt = cg * l_out;
    const int act = CDX_RL(w, CDX_W_ACT), coff = CDX_RL(w, CDX_W_DST_COFF);
    const float oscale = (flags & CDX_F_SCALE) ? __int_as_float(CDX_RL(w, CDX_W_SCALE)) : 1.0f;
    const bool gn = flags & CDX_F_GN_MISH;
    const bool col_norm = CDX_RL(w, CDX_W_NORM) == CDX_NORM_COLUMN;
    const bool gn_fast = gn && sh >= 0 && cnt <= 64 * CDX_EPI_REGS && groups <= CDX_N_WAVES;
    if (gn_fast) {
        // one wave per group; a lane keeps its <= CDX_EPI_REGS elements in registers across the three passes
        const float* __restrict__ gamma = wblob + CDX_RL(w, CDX_W_GAMMA);
        const float* __restrict__ beta = wblob + CDX_RL(w, CDX_W_BETA);
        float bi[CDX_EPI_REGS], ga[CDX_EPI_REGS], be[CDX_EPI_REGS];
        int nn[CDX_EPI_REGS], cc[CDX_EPI_REGS];
        const bool active = wave < groups;
        if (active) {
#pragma unroll
            for (int t = 0; t < CDX_EPI_REGS; ++t) {
                const int e = min(lane + 64 * t, cnt - 1);
                nn[t] = e >> sh;
                cc[t] = wave * cg + (e & (cg - 1));
                bi[t] = bias[cc[t]]; ga[t] = gamma[cc[t]]; be[t] = beta[cc[t]];
            }
        }
        stamp(prof ? prof + 1 : nullptr, tid);
        __syncthreads();
        stamp(prof ? prof + 2 : nullptr, tid);
        if (active) {
            const float inv_cnt = __int_as_float(CDX_RL(w, CDX_W_INV_CNT));
            float v[CDX_EPI_REGS];
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < CDX_EPI_REGS; ++t) {
                float acc = bi[t];
                for (int ks = 0; ks < ksplit; ++ks) acc += lds[scratch + (ks * l_out + nn[t]) * sstride + cc[t]];
                v[t] = acc;
                s += (lane + 64 * t < cnt) ? acc : 0.f;
            }
            const float mean = wave_sum(s) * inv_cnt;
            float s2 = 0.f;
#pragma unroll
            for (int t = 0; t < CDX_EPI_REGS; ++t) {
                const float d = v[t] - mean;
                s2 += (lane + 64 * t < cnt) ? d * d : 0.f;
            }
            const float rstd = __builtin_amdgcn_rsqf(wave_sum(s2) * inv_cnt + CDX_GN_EPS);
#pragma unroll
            for (int t = 0; t < CDX_EPI_REGS; ++t) {
                if (lane + 64 * t < cnt) {
                    float y = act_f<FULL>((v[t] - mean) * rstd * ga[t] + be[t], act);
                    if (flags & CDX_F_ADD_EMB) y += lds[emb + cc[t]];
                    if (flags & CDX_F_FILM) y = y * lds[emb + cc[t]] + lds[emb + c_out + cc[t]];
                    if (flags & CDX_F_ADD_RES) y += lds[res + (nn[t] + CDX_HALO) * rstride + cc[t]];
                    lds[dst + (nn[t] + CDX_HALO) * dstride + coff + cc[t]] = y * oscale;
                }
            }
        }
    } else if (gn) {
        // general GroupNorm path (odd group sizes / long horizons): three passes through LDS
        const float* __restrict__ gamma = wblob + CDX_RL(w, CDX_W_GAMMA);
        const float* __restrict__ beta = wblob + CDX_RL(w, CDX_W_BETA);
        const float inv_cnt = __int_as_float(CDX_RL(w, CDX_W_INV_CNT));
        const float inv_cg = 1.0f / (float)cg;
        stamp(prof ? prof + 1 : nullptr, tid);
        __syncthreads();
        stamp(prof ? prof + 2 : nullptr, tid);
        for (int gi = wave; gi < groups; gi += CDX_N_WAVES) {
            float s = 0.f;
            for (int e = lane; e < cnt; e += 64) {
                const int n = div_small(e, cg, inv_cg), c = gi * cg + (e - n * cg);
                float v = bias[c];
                for (int ks = 0; ks < ksplit; ++ks) v += lds[scratch + (ks * l_out + n) * sstride + c];
                lds[scratch + n * sstride + c] = v;  // owned by this lane only
                s += v;
            }
            const float mean = wave_sum(s) * inv_cnt;
            float s2 = 0.f;
            for (int e = lane; e < cnt; e += 64) {
                const int n = div_small(e, cg, inv_cg), c = gi * cg + (e - n * cg);
                const float d = lds[scratch + n * sstride + c] - mean;
                s2 += d * d;
            }
            const float rstd = __builtin_amdgcn_rsqf(wave_sum(s2) * inv_cnt + CDX_GN_EPS);
            for (int e = lane; e < cnt; e += 64) {
                const int n = div_small(e, cg, inv_cg), c = gi * cg + (e - n * cg);
                float v = act_f<FULL>((lds[scratch + n * sstride + c] - mean) * rstd * gamma[c] + beta[c], act);
                if (flags & CDX_F_ADD_EMB) v += lds[emb + c];
                if (flags & CDX_F_FILM) v = v * lds[emb + c] + lds[emb + c_out + c];
                if (flags & CDX_F_ADD_RES) v += lds[res + (n + CDX_HALO) * rstride + c];
                lds[dst + (n + CDX_HALO) * dstride + coff + c] = v * oscale;
            }
        }
    } else if (FULL && col_norm) {
        // per-column normalisation (GroupNorm1d on (b, C) / LayerNorm of the MLP backbones): statistics over the
        // cg channels of ONE column (= one sample of the batch tile).  Three short passes through LDS -- deliberately compact
        // code: these layers are tiny and the instruction cache is what the U-Net path is sensitive to.
        const float* __restrict__ gamma = wblob + CDX_RL(w, CDX_W_GAMMA);
        const float* __restrict__ beta = wblob + CDX_RL(w, CDX_W_BETA);
        const float inv_cg = __int_as_float(CDX_RL(w, CDX_W_INV_CNT));
        const float inv_groups = 1.0f / (float)groups;
        stamp(prof ? prof + 1 : nullptr, tid);
        __syncthreads();
        stamp(prof ? prof + 2 : nullptr, tid);
        // A (column, group) pair is cg channels of one sample: far too little for a wave (the wave-per-pair loop spent 28 k cycles
        // per 256-wide layer of PearceMlp, twice its K loop).  `seg` adjacent lanes own a pair instead -- the largest power of two
        // that still keeps all CDX_THREADS lanes busy -- and reduce with xor shuffles inside their segment.
        const int pairs = l_out * groups;
        int seg = 64;
        while (seg > 1 && (seg > cg || pairs * (seg >> 1) >= CDX_THREADS)) seg >>= 1;
        const int sl = tid & (seg - 1), slot = tid / seg, per_round = CDX_THREADS / seg;
        for (int p0 = 0; p0 < pairs; p0 += per_round) {
            const bool live = p0 + slot < pairs;                 // dead segments shadow the last pair (uniform shuffles), never store
            const int pair = live ? p0 + slot : pairs - 1;
            const int n = div_small(pair, groups, inv_groups), c0 = (pair - n * groups) * cg;
            float s = 0.f;
            for (int e = sl; e < cg; e += seg) {
                const int c = c0 + e;
                float v = bias[c];
                for (int ks = 0; ks < ksplit; ++ks) v += lds[scratch + (ks * l_out + n) * sstride + c];
                if (live) lds[scratch + n * sstride + c] = v;    // owned by this lane only
                s += v;
            }
            for (int o = seg >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * inv_cg;
            float s2 = 0.f;
            if (live)
                for (int e = sl; e < cg; e += seg) {
                    const float d = lds[scratch + n * sstride + c0 + e] - mean;
                    s2 += d * d;
                }
            for (int o = seg >> 1; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
            const float rstd = __builtin_amdgcn_rsqf(s2 * inv_cg + CDX_GN_EPS);
            if (live)
                for (int e = sl; e < cg; e += seg) {
                    const int c = c0 + e;
                    float y = act_f<FULL>((lds[scratch + n * sstride + c] - mean) * rstd * gamma[c] + beta[c], act);
                    if (flags & CDX_F_ADD_EMB) y += lds[emb + c];
                    if (flags & CDX_F_ADD_RES) y += lds[res + (n + CDX_HALO) * rstride + c];
                    lds[dst + (n + CDX_HALO) * dstride + coff + c] = y * oscale;
                }
        }
    } else {
        // plain conv (down/up-sample, 1x1 residual, output head): bias (+ residual / accumulate)
        const float inv_cout = __int_as_float(CDX_RL(w, CDX_W_INV_COUT));
        const int total = c_out * l_out;
        int n0, c0;
        float b0;
        {
            const int e = min(tid, total - 1);
            n0 = div_small(e, c_out, inv_cout); c0 = e - n0 * c_out; b0 = bias[c0];
        }
        stamp(prof ? prof + 1 : nullptr, tid);
        __syncthreads();
        stamp(prof ? prof + 2 : nullptr, tid);
        for (int e = tid; e < total; e += CDX_THREADS) {
            int n = n0, c = c0;
            float v = b0;
            if (e != tid) { n = div_small(e, c_out, inv_cout); c = e - n * c_out; v = bias[c]; }
            for (int ks = 0; ks < ksplit; ++ks) v += lds[scratch + (ks * l_out + n) * sstride + c];
            v = act_f<FULL>(v, act);
            if (flags & CDX_F_ADD_EMB) v += lds[emb + c];
            if (flags & CDX_F_ADD_RES) v += lds[res + (n + CDX_HALO) * rstride + c];
            v *= oscale;
            const int o = dst + (n + CDX_HALO) * dstride + coff + c;
            if (flags & CDX_F_ACCUM) v += lds[o];
            lds[o] = v;
        }
    }
    __syncthreads();
}

template <bool FULL, bool PROF>
__device__ __forceinline__ void run_program(const cdx_unet1d_launch& L, float* __restrict__ lds, int step, int branch, bool use_cond,
                            int b, int tid, Prefetch& pre) {
    // (PROF = false: `pslot` below is a compile-time null and every stamp and its scalar branch folds away -- 1-3 % on the v2 kernel)
    const bool profiling = PROF && L.prof != nullptr && b == 0 && step == 0 && branch == 0;
    unsigned long long* lprof = reinterpret_cast<unsigned long long*>(lds + L.prof_off);
    const int* __restrict__ ldsi = reinterpret_cast<const int*>(lds);
    const int lane = tid & 63;
    const int dlane = lane < CDX_OP_WORDS ? lane : 0;
    // work-item tables: in LDS behind the ops when they fit, else read from the global copy (large programs: the
    // ~1 us per item is noise next to their K loops).  Generic pointer -> flat loads either way.
    const int* __restrict__ itab = L.items_in_lds ? ldsi + L.desc_off : L.ops;
    int wn = ldsi[L.desc_off + dlane];                       // descriptor of op 0, one word per lane
    for (int oi = 0; oi < L.n_ops; ++oi) {
        const int w = wn;
        const int nxt = oi + 1 < L.n_ops ? oi + 1 : 0;
        wn = ldsi[L.desc_off + nxt * CDX_OP_WORDS + dlane];  // next descriptor: its latency hides behind this op
        const int kind = CDX_RL(w, CDX_W_KIND);
        const int op[9] = {kind, CDX_RL(w, 1), CDX_RL(w, 2), CDX_RL(w, 3), CDX_RL(w, 4), CDX_RL(w, 5), CDX_RL(w, 6),
                           CDX_RL(w, 7), CDX_RL(w, 8)};      // linear / flatten / temb ops only use words 0..8
        unsigned long long* pslot = profiling ? lprof + (size_t)oi * 8 : nullptr;
        stamp(pslot, tid);
        if (kind == CDX_OP_CONV) {
            conv_op<FULL>(w, oi + 1 < L.n_ops ? wn : 0, itab, L.wblob, lds, L.scratch_off, L.zrow_off,
                          branch * L.pred_branch_floats, tid, pre, pslot);
        } else if (kind == CDX_OP_LINEAR) {
            const int n_in = op[CDX_L_NIN], n_out = op[CDX_L_NOUT];
            const float* __restrict__ w = L.wblob + op[CDX_L_WOFF];   // [n_in][n_out]
            const float* __restrict__ bb = L.wblob + op[CDX_L_BOFF];
            const int src = op[CDX_L_SRC], dst = op[CDX_L_DST];
            const bool post = op[CDX_L_FLAGS] & CDX_F_POST_MISH;
            const bool raw_copy = op[CDX_L_FLAGS] & CDX_F_RAW_COPY;
            const int dst2 = op[CDX_L_DST2];
            int kparts = CDX_THREADS / n_out;
            kparts = kparts > 16 ? 16 : kparts;
            if (kparts > 1) {
                // narrow output: (k-part, output) pairs across the workgroup, partials through the scratch area
                if (tid < n_out * kparts) {
                    const int part = tid / n_out, o = tid - part * n_out;
                    const int i0 = part * n_in / kparts, i1 = (part + 1) * n_in / kparts;
                    float acc = 0.f;
#pragma unroll 8
                    for (int i = i0; i < i1; ++i) acc = fmaf(w[(size_t)i * n_out + o], lds[src + i], acc);
                    lds[L.scratch_off + part * n_out + o] = acc;
                }
                __syncthreads();
                if (tid < n_out) {
                    float acc = bb[tid];
                    for (int part = 0; part < kparts; ++part) acc += lds[L.scratch_off + part * n_out + tid];
                    if (raw_copy) lds[dst2 + tid] = acc;
                    lds[dst + tid] = post ? mish_f(acc) : acc;
                }
            } else {
                // wide output: 4 outputs per thread so 4 x unroll independent weight loads are in flight
                for (int o0 = tid; o0 < n_out; o0 += CDX_THREADS * 4) {
                    int oo[4];
                    float acc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int o = o0 + u * CDX_THREADS;
                        oo[u] = o < n_out ? o : o0;
                        acc[u] = bb[oo[u]];
                    }
#pragma unroll 4
                    for (int i = 0; i < n_in; ++i) {
                        const float xi = lds[src + i];
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[u] = fmaf(w[(size_t)i * n_out + oo[u]], xi, acc[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (o0 + u * CDX_THREADS < n_out) {
                            if (raw_copy) lds[dst2 + oo[u]] = acc[u];
                            lds[dst + oo[u]] = post ? mish_f(acc[u]) : acc[u];
                        }
                }
            }
            __syncthreads();
        } else if (FULL && kind == CDX_OP_FILL) {
            // broadcast a vector into a channel range of every row of a slot (batch-invariant features of MLP tiles)
            const int n = op[CDX_L_NIN], rows = op[CDX_L_NOUT], src = op[CDX_L_SRC], dst = op[CDX_L_DST];
            const int sstr = op[CDX_L_WOFF], fcoff = op[CDX_L_BOFF];
            for (int e = tid; e < n * rows; e += CDX_THREADS) {
                const int r = e / n, i = e - r * n;
                lds[dst + (r + CDX_HALO) * sstr + fcoff + i] = lds[src + i];
            }
            __syncthreads();
        } else if (kind == CDX_OP_FLATTEN) {
            // slot (channel-last rows) -> vector in torch's (C, L) flatten order: v[c*L + l] = slot[l][c]
            const int C = op[CDX_L_NIN], Lp = op[CDX_L_NOUT], src = op[CDX_L_SRC], dst = op[CDX_L_DST];
            const int sstr = op[CDX_L_WOFF];
            for (int c = tid; c < C; c += CDX_THREADS)
                for (int l = 0; l < Lp; ++l) lds[dst + c * Lp + l] = lds[src + (l + CDX_HALO) * sstr + c];
            __syncthreads();
        } else if (kind == CDX_OP_LOAD_COND) {
            // raw per-trajectory condition features -> vec (zeros when the launch carries no condition / uncond branch)
            const int n = op[CDX_L_NIN], dst = op[CDX_L_DST];
            for (int i = tid; i < n; i += CDX_THREADS)
                lds[dst + i] = use_cond ? L.cond[(size_t)b * L.cond_dim + i] : 0.f;
            __syncthreads();
        } else {  // CDX_OP_LOAD_TEMB
            const int n = op[CDX_L_NIN], dst = op[CDX_L_DST];
            for (int i = tid; i < n; i += CDX_THREADS) {
                float v = L.temb[(size_t)(L.temb_per_sample ? b : step) * L.emb_dim + i];
                if (use_cond && L.tile == 0 && L.cond_dim == 0) v += L.cond[(size_t)b * L.emb_dim + i];
                lds[dst + i] = v;
            }
            __syncthreads();
        }
        stamp(pslot ? pslot + 3 : nullptr, tid);
    }
}

template <bool FULL, bool PROF>
__global__ __launch_bounds__(CDX_THREADS) void cdx_unet1d_kernel(const cdx_unet1d_launch L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int H = L.horizon, D = L.dim, HD = H * D;
    const size_t xbase = (size_t)b * HD;
    unsigned long long* lprof = reinterpret_cast<unsigned long long*>(lds + L.prof_off);
    if (PROF && L.prof && b == 0) stamp(lprof + (size_t)L.n_ops * 8, tid);

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

    // ---- further kernel-lifetime slots (MLP context): clear once, then (tile programs) load the per-sample
    //      condition features into their channel range; an absent / unused condition stays zero ----
    for (int i = tid * 4; i < L.zero_floats; i += CDX_THREADS * 4)
        *reinterpret_cast<float4*>(lds + L.zero_off + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (L.tile > 0 && L.cond != nullptr && L.cfg_mode == 1) {
        for (int e = tid; e < H * L.cond_dim; e += CDX_THREADS) {
            const int n = e / L.cond_dim, i = e - n * L.cond_dim;
            lds[L.cond_slot_off + (n + CDX_HALO) * L.cond_slot_stride + L.cond_coff + i] =
                L.cond[((size_t)b * H + n) * L.cond_dim + i];
        }
    }
    __syncthreads();

    // ---- layer descriptors + work-item tables: global -> LDS once (scalar-cache misses cost ~800 cycles per op) ----
    {
        int* ldsi = reinterpret_cast<int*>(lds);
        for (int i = tid; i < L.desc_words; i += CDX_THREADS) ldsi[L.desc_off + i] = L.ops[i];
    }
    __syncthreads();
    Prefetch pre;
    pre.ok = 0;

    // EDM plans (kinds 5/6; a plan is all-EDM or not at all): the network sees c_in * x, so the authoritative state lives in a
    // dense side buffer and the state slot is rewritten with the scaled copy before every forward.
    const int PV = (HD + 3) & ~3;                        // prev | x_old | x_true, PV floats each (program.py: prev region)
    const bool edm = L.n_steps > 0 && L.steps[0].kind >= 5;
    if (edm) {
        for (int e = tid; e < HD; e += CDX_THREADS) {
            const int n = e / D, c = e - n * D;
            lds[L.prev_off + 2 * PV + e] = lds[L.x_off + (n + CDX_HALO) * L.x_stride + c];
        }
        __syncthreads();
    }

    const int n_iter = L.n_steps > 0 ? L.n_steps : 1;
    for (int step = 0; step < n_iter; ++step) {
        if (edm) {
            const float c_in = L.steps[step].alpha;
            for (int e = tid; e < HD; e += CDX_THREADS) {
                const int n = e / D, c = e - n * D;
                lds[L.x_off + (n + CDX_HALO) * L.x_stride + c] = c_in * lds[L.prev_off + 2 * PV + e];
            }
            __syncthreads();
        }
        const int n_branch = (L.cfg_mode == 2) ? 2 : 1;
        for (int br = 0; br < n_branch; ++br) {
            const bool use_cond = (L.cond != nullptr) && (L.cfg_mode == 1 || (L.cfg_mode == 2 && br == 0));
            if (L.tile > 0 && L.cfg_mode == 2) {
                // tile programs keep the condition features in a kernel-lifetime context slot: the CFG pair (reference
                // diffusionsde.py:185-199, cond | zeros) rewrites that channel range before each branch
                for (int e = tid; e < H * L.cond_dim; e += CDX_THREADS) {
                    const int n = e / L.cond_dim, i = e - n * L.cond_dim;
                    lds[L.cond_slot_off + (n + CDX_HALO) * L.cond_slot_stride + L.cond_coff + i] =
                        use_cond ? L.cond[((size_t)b * H + n) * L.cond_dim + i] : 0.f;
                }
                __syncthreads();
            }
            run_program<FULL, PROF>(L, lds, step, br, use_cond, b, tid, pre);
        }
        if (L.n_steps == 0 && L.out_vec_len > 0) {  // forward-only, vector head (classifier): emit the head output
            for (int i = tid; i < L.out_vec_len; i += CDX_THREADS)
                L.x_out[(size_t)b * L.out_vec_len + i] = lds[L.out_vec_off + i];
            return;
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
            const float x = edm ? lds[L.prev_off + 2 * PV + e] : lds[xo];
            float p = lds[po];
            if (L.cfg_mode == 2) p = L.cfg_w * p + (1.0f - L.cfg_w) * lds[po + L.pred_branch_floats];
            float xn;
            if (st.kind >= 5) {
                // EDM (newedm.py:387-401, legacy edm.py:118-160): D = clip(c_skip x + c_out F), slope = (x - D) / sigma
                float dn = k0 * x + k1 * p;
                if (L.x_min) dn = fmaxf(dn, L.x_min[e]);
                if (L.x_max) dn = fminf(dn, L.x_max[e]);
                if (st.kind == 7) {                      // consistency model: x <- f(x) [mask], then re-noise for the next level
                    xn = dn;
                    if (L.fix_mask) {
                        const float m = L.fix_mask[e];
                        xn = xn * (1.0f - m) + L.prior[xbase + e] * m;
                    }
                    if (st.noise_idx >= 0) xn += k3 * L.noise[((size_t)st.noise_idx * L.batch + b) * HD + e];
                    lds[L.prev_off + 2 * PV + e] = xn;
                    continue;
                }
                const float sl = (x - dn) / k2;
                if (st.kind == 5) {
                    xn = x - sl * k3;
                    if (st.push) { lds[L.prev_off + e] = sl; lds[L.prev_off + PV + e] = x; }
                } else {
                    xn = lds[L.prev_off + PV + e] - (lds[L.prev_off + e] + sl) / 2.0f * k3;
                }
                if (L.fix_mask) {
                    const float m = L.fix_mask[e];
                    xn = xn * (1.0f - m) + L.prior[xbase + e] * m;
                }
                lds[L.prev_off + 2 * PV + e] = xn;
                continue;
            }
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
            if (st.kind >= 3) {
                // legacy DDPM class (reference diffusion/ddpm.py:153-164, 230-241): the fix-mask is applied to the
                // *prediction* (eps: pred*(1-m); x0: pred*(1-m) + x*m), then the ancestral posterior mean
                const float m = L.fix_mask ? L.fix_mask[e] : 0.f;
                if (st.kind == 3) {
                    p = p * (1.0f - m);
                    xn = k0 * (x - k1 * p);
                } else {
                    p = p * (1.0f - m) + x * m;
                    xn = k0 * (k1 * x + k2 * p);
                }
                if (st.noise_idx >= 0) xn += k3 * L.noise[((size_t)st.noise_idx * L.batch + b) * HD + e];
            } else if (st.kind == 0) {
                xn = k0 * (x - k1 * eps) + k2 * eps;
                if (st.noise_idx >= 0) xn += k3 * L.noise[((size_t)st.noise_idx * L.batch + b) * HD + e];
            } else if (st.kind == 1) {
                xn = k0 * ((x - k1 * eps) / k2) + k3 * eps;
            } else {
                if (st.flags & CDX_STEP_MASK_PRED) {     // legacy DPMSolver (dpmsolver.py:257-264): mask on the prediction
                    const float m = L.fix_mask ? L.fix_mask[e] : 0.f;
                    eps = eps * (1.0f - m);
                    xth = xth * (1.0f - m) + x * m;
                }
                float v = (st.vsel & 1) ? xth : eps;     // 0 eps, 1 x_theta, 2 multistep on x_theta, 3 multistep on eps
                if (st.vsel == 2) v = k3 * xth - k4 * lds[L.prev_off + e];
                if (st.vsel == 3) v = k3 * eps - k4 * lds[L.prev_off + e];
                xn = k0 * x - k1 * v;
                if (st.noise_idx >= 0) xn += k2 * L.noise[((size_t)st.noise_idx * L.batch + b) * HD + e];
            }
            if (L.fix_mask) {
                const float m = L.fix_mask[e];
                xn = xn * (1.0f - m) + L.prior[xbase + e] * m;
            }
            if (st.push) lds[L.prev_off + e] = st.push == 2 ? eps : xth;
            lds[xo] = xn;
        }
        __syncthreads();
    }
    for (int e = tid; e < HD; e += CDX_THREADS) {
        const int n = e / D, c = e - n * D;
        L.x_out[xbase + e] = edm ? lds[L.prev_off + 2 * PV + e] : lds[L.x_off + (n + CDX_HALO) * L.x_stride + c];
    }
    if (PROF && L.prof && b == 0) {
        stamp(lprof + (size_t)L.n_ops * 8 + 1, tid);
        __syncthreads();
        for (int i = tid; i < L.n_ops * 8 + 2; i += CDX_THREADS) L.prof[i] = lprof[i];
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
    if (L->out_vec_len > 0 && L->n_steps != 0) { set_err("vector-output programs run in forward mode only"); return CDX_EINVAL; }
    if (L->n_steps > 0 && L->pred_stride == 0) { set_err("sampling needs a program with a prediction slot"); return CDX_EINVAL; }
    if (L->fix_mask && !L->prior) { set_err("fix_mask given without prior"); return CDX_EINVAL; }
    if (L->cfg_mode < 0 || L->cfg_mode > 2) { set_err("cfg_mode must be 0, 1 or 2"); return CDX_EINVAL; }
    if (L->cfg_mode == 2 && !L->cond) { set_err("cfg_mode 2 needs cond"); return CDX_EINVAL; }
    if (L->tile > 0 && L->temb_per_sample) { set_err("tile programs: per-step timesteps only"); return CDX_EINVAL; }
    if (L->tile > 0 && L->tile != L->horizon) { set_err("tile programs: horizon must equal the tile size"); return CDX_EINVAL; }
    if ((L->zero_off | L->zero_floats) & 3) { set_err("zero range must be 16-byte aligned"); return CDX_EINVAL; }
    if ((L->x_off | L->pred_off | L->prev_off | L->scratch_off | L->x_stride | L->pred_stride | L->pred_branch_floats) & 3) {
        set_err("LDS offsets/strides must be multiples of 4 floats"); return CDX_EINVAL;
    }
    if (L->desc_words < L->n_ops * CDX_OP_WORDS || L->desc_off < 0 || L->desc_off + L->desc_words > L->lds_floats) {
        set_err("descriptor area does not fit the LDS plan"); return CDX_EINVAL;
    }
    if (L->prof && (L->prof_off <= 0 || (L->prof_off & 1) || L->prof_off + 2 * (L->n_ops * 8 + 2) > L->lds_floats)) {
        set_err("profiling requested but the LDS plan has no stamp area"); return CDX_EINVAL;
    }
    const size_t lds_bytes = (size_t)L->lds_floats * sizeof(float);
    if (lds_bytes > 160u * 1024u) { set_err("program needs more than 160 KiB of LDS"); return CDX_ELDS; }
    // two instantiations: the lean U-Net kernel and the full-featured one for batch-tiled MLP programs
    auto kern = L->prof ? (L->tile > 0 ? cdx_unet1d_kernel<true, true> : cdx_unet1d_kernel<false, true>)
                        : (L->tile > 0 ? cdx_unet1d_kernel<true, false> : cdx_unet1d_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return CDX_EHIP; }
    hipLaunchKernelGGL(kern, dim3(L->batch), dim3(CDX_THREADS), lds_bytes,
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
