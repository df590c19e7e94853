// cdx_unet2.hip -- second-generation fused "program" kernel for 1-D temporal U-Net denoisers on gfx950 (MI355X / CDNA4).
//
// Replaces the ~170 ATen launches per denoiser forward plus the solver arithmetic of the reference loop
// (cleandiffuser/diffusion/diffusionsde.py:526-594 driving cleandiffuser/nn_diffusion/jannerunet.py:154-201) with ONE launch per
// sample() call: one workgroup per trajectory (or per 2 / 3 trajectories), every activation in LDS, the weights streamed from L2
// as 1-KiB MFMA records.  The kernel as it stands (what each choice bought is in DESIGN.md section 3a):
//
//   * 8 wave64 per workgroup (two per SIMD: a lone wave reaches about half the MFMA issue rate); a conv's output is cut into (row
//     tile x column group x K slice) work items, one per wave; each wave keeps an 8-deep register ring of weight records in flight
//     with counted s_waitcnt, the head of the NEXT op's stream is issued right after this op's K loop.  A 4-wave shape (16-deep
//     ring) still exists for LDS plans that only fit two trajectories that way.
//   * an op descriptor is ONE coalesced dword load per wave (lane k = word k); fields are decoded with constant-lane v_readlane --
//     the scalar-load version spent ~900 cycles per op on dependent s_loads and SGPR spills;
//   * activation slots carry two zero halo rows, so a work item's B-operand address is linear in (tap, chunk): per record the K loop
//     issues {wait, 4 MFMAs per (trajectory, column tile), one ds_read_b128, one buffer_load}; ConvTranspose1d(4,2,1) is two 2-tap
//     convs (one per output parity); the ResidualBlock's 1x1 skip conv rides as extra work items of the block's second conv;
//   * FiLM vectors come from a per-(step[, trajectory]) table built by cdx_unet2_embtab_kernel (the embedding MLP is not part of the
//     per-forward program);
//   * epilogue: 32 lanes per GroupNorm group, one float4 of consecutive channels per lane, statistics in ONE shifted pass (sums of
//     x - s and (x - s)^2, half-wave DPP reductions), Mish, + FiLM, + residual, one float4 LDS store; with T >= 2 waves 0-3 / 4-7
//     run the trajectories' epilogues side by side;
//   * the next op's epilogue parameters are fetched behind this op's staging barrier and the descriptor after next right after the
//     K loop (OpFetch): an op starts with nothing recent in the vector-memory queue;
//   * T = 1, 2 or 3 trajectories per workgroup share every streamed record (3: compact programs -- state and multistep memory in
//     global memory); BWD instantiations add the classifier's backward-data ops (guided sampling in the same launch); COND
//     instantiations add per-trajectory FiLM rows, the classifier-free-guidance pair and the EDM / consistency step kinds.
//
// Executable specification / CPU twin: oracle/lane_sim2.py.  Program format: engine/program2.py, csrc/cdx_ops2.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/cdx.h"
#include "cdx_ops2.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Descriptor / item words are read through the constant address space: a wave-uniform address there is an s_load (scalar
// cache, SGPR destination) -- no vector load + v_readfirstlane decode.
typedef const int __attribute__((address_space(4))) cint;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
static __device__ __forceinline__ const cint* as_const(const int* p) { return (const cint*)p; }
#pragma clang diagnostic pop

void cdx_set_err(const char* msg);          // cdx_common.hip

// Workgroup shapes (template parameter NWV = wave64 per workgroup):
//   4 waves, one per SIMD, 16 weight records in flight per wave -- up to 512 VGPRs per lane;
//   8 waves, two per SIMD,  8 weight records in flight per wave -- 256 VGPRs per lane.  A lone wave on a SIMD issues about one
//     instruction per four clocks and cannot keep the matrix pipe busy across its own ds_read / s_waitcnt bubbles (round-2 op
//     profile: ~62 clk per 16x16x4 MFMA, 15-22 per 4x4x1, i.e. half rate); a second wave fills those slots.  With two
//     trajectories per workgroup waves 0-3 / 4-7 also run the two epilogues side by side.
template <int NWV> struct WG {
    static constexpr int THREADS = NWV * 64;
    static constexpr int PF = NWV == 8 ? 8 : 16;                                  // weight ring depth (1-KiB records per wave)
                                                                                  // (16 at 8 waves: 10 % slower, measured again in round 2)
    static constexpr int OPW = CDX2_HDR_WORDS + NWV * CDX2_ITEM_WORDS;             // words per op descriptor
};

namespace {

__device__ __forceinline__ float mish2(float x) {
    // x * tanh(softplus(x)), tanh(log(1+e^x)) = n / (n + 2), n = e^x (e^x + 2); softplus threshold 20 as ATen
    const float e = __expf(fminf(x, 20.0f));
    const float n = e * (e + 2.0f);
    return x * n * __builtin_amdgcn_rcpf(n + 2.0f);      // (x > 20: e = exp(20), n / (n + 2) rounds to 1.0f -- no select needed)
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, moved);
}
// sum over the 32 lanes of this lane's half-wave (one GroupNorm group): DPP inside the 16-lane rows, then the two rows of the
// half are combined through SGPRs
__device__ __forceinline__ float half_sum(float v, int lane) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror -> 16-lane row sums
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return lane < 32 ? r0 + r1 : r2 + r3;
}

// sum over the aligned group of 2^w_log2 lanes this lane belongs to (w_log2 <= 5, wave-uniform): the per-sample GroupNorm of the
// batch-tiled MLP programs reduces over the float4 items of ONE position
__device__ __forceinline__ float seg_sum(float v, int w_log2, int lane) {
    if (w_log2 >= 1) v = dpp_add<0xB1>(v);
    if (w_log2 >= 2) v = dpp_add<0x4E>(v);
    if (w_log2 >= 3) v = dpp_add<0x141>(v);
    if (w_log2 >= 4) v = dpp_add<0x140>(v);
    if (w_log2 >= 5) {
        const int iv = __builtin_bit_cast(int, v);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
        v = lane < 32 ? r0 + r1 : r2 + r3;
    }
    return v;
}

// Activations of the batch-tiled MLP programs: id = CDX_ACT_* (include/cdx.h) + 1, wave-uniform (a scalar branch)
__device__ __forceinline__ float act2_f(float x, int id) {
    switch (id) {
        case 2: return mish2(x);                                         // CDX_ACT_MISH
        case 3: {                                                        // CDX_ACT_GELU_ERF: erf by Abramowitz-Stegun 7.1.26, |err| < 1.5e-7
            const float z = fabsf(x) * 0.70710678118654752f;
            const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
            const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
            return 0.5f * x * (1.0f + copysignf(1.0f - poly * __expf(-z * z), x));
        }
        case 4: return x > 0.f ? x : 0.01f * x;                          // CDX_ACT_LEAKY
        case 5: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));     // CDX_ACT_SILU
        case 6: return fmaxf(x, 0.f);                                    // CDX_ACT_RELU
        case 7: return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));   // CDX_ACT_GELU_TANH
        case 9: return tanhf(x);                                         // CDX_ACT_TANH
        default: return x;                                               // 1 = CDX_ACT_NONE
    }
}

struct M16 {   // v_mfma_f32_16x16x4_f32: 16 rows x 16 cols, a record = 16 K values
    static constexpr int COLS = 16, KSTEP = 16;
    static __device__ __forceinline__ int col(int lane) { return lane & 15; }
    static __device__ __forceinline__ int koff(int lane) { return 4 * (lane >> 4); }
    static __device__ __forceinline__ int drow(int lane) { return 4 * (lane >> 4); }
    static __device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
struct M4 {    // v_mfma_f32_4x4x1_16b_f32: 16 blocks of 4x4 = 64 rows x 4 cols, a record = 4 K values
    static constexpr int COLS = 4, KSTEP = 4;
    static __device__ __forceinline__ int col(int lane) { return lane & 3; }
    static __device__ __forceinline__ int koff(int) { return 0; }
    static __device__ __forceinline__ int drow(int lane) { return 4 * (lane >> 2); }
    static __device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    }
};

typedef const cdx_unet2_launch __attribute__((address_space(4))) KArg;      // the launch struct as it lies in the kernarg segment

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
static __device__ __forceinline__ const KArg* kernarg() { return (const KArg*)__builtin_amdgcn_kernarg_segment_ptr(); }
#pragma clang diagnostic pop

struct Geom {                    // wave-uniform description of one conv op's K loop
    int l_cols, cstride, ostride, sstride, stage;
    int gsh, grows;              // grouped ops (GRP kernels): column m = trajectory (m >> gsh) of the group, position m & (2^gsh - 1);
                                 // the trajectories' sub-slots lie `grows` rows apart.  grows == 0: an ordinary op
};

// One wave's job: (row tile, column group, K slice [, output parity of a transposed conv]) over ONE source slot.
struct Item { int woff, nq, tap, cc, part, col0, pad, ooff, src, sstr, ccn; };

__device__ __forceinline__ Item make_item(int woff, int nq, int tc, int part, int col0, int po, int ss, int ccn) {
    return Item{woff, nq, tc & 255, tc >> 8, part, col0, po & 255, po >> 8, ss & 0xffff, ss >> 16, ccn};
}
__device__ __forceinline__ Item load_item(const cint* it) {        // items past the inline four: scalar loads (rare)
    return make_item(it[CDX2_I2_WOFF], it[CDX2_I2_NQ], it[CDX2_I2_TAPCC], it[CDX2_I2_PART], it[CDX2_I2_COL0],
                     it[CDX2_I2_PADOOFF], it[CDX2_I2_SRCSTR], it[CDX2_I2_CCN]);
}

// An op descriptor is read with ONE dword load per wave: lane k holds word k, fields come out with v_readlane.
// (The scalar-load version of this kernel spent ~900 cycles per op waiting on dependent s_loads / spilling their results.)
#define CDX2_DW(vd, k) __builtin_amdgcn_readlane((vd), (k))
template <int NWV>
__device__ __forceinline__ int load_desc(const int* __restrict__ ops, int op, int lane, int wave) {
    const int w = lane < CDX2_HDR_WORDS ? lane : CDX2_HDR_WORDS + wave * CDX2_ITEM_WORDS + (lane & (CDX2_ITEM_WORDS - 1));
    return ops[op * WG<NWV>::OPW + w];
}
// A wave's view `vd` of an op descriptor: lanes 0-31 hold the header words, lanes 32-39 the words of THIS wave's inline item
// (item `wave` of the op) -- see load_desc(); every field comes out with a constant-lane v_readlane.
__device__ __forceinline__ Item inline_item(int vd) {
    const int b = CDX2_HDR_WORDS;
    return make_item(CDX2_DW(vd, b + CDX2_I2_WOFF), CDX2_DW(vd, b + CDX2_I2_NQ), CDX2_DW(vd, b + CDX2_I2_TAPCC),
                     CDX2_DW(vd, b + CDX2_I2_PART), CDX2_DW(vd, b + CDX2_I2_COL0), CDX2_DW(vd, b + CDX2_I2_PADOOFF),
                     CDX2_DW(vd, b + CDX2_I2_SRCSTR), CDX2_DW(vd, b + CDX2_I2_CCN));
}

// Profiling stamps go to LDS (a global store would be waited on by the next vmcnt wait and distort the phase being measured).
__device__ __forceinline__ void stamp(unsigned long long* slot, int tid) {
    if (slot && tid == 0) *slot = __builtin_amdgcn_s_memtime();
}

// ---- weight records: loaded through a raw buffer descriptor (uniform byte offset in an SGPR, lane * 16 in one VGPR -- no 64-bit
// vector address arithmetic per load: +2.4 % at two trajectories per workgroup) with a compile-time cache policy (CDX2_WPOLICY:
// 0 default, 1 sc0, 2 nt, 16 sc1).  Measured on MI355X: sc0 / sc1 change nothing; nt is 30-40 % SLOWER -- every CU of an XCD streams
// the same records, and non-temporal lines do not stay in the L2 for the other 31.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef CDX2_WPOLICY
#define CDX2_WPOLICY 0
#endif
#ifndef CDX2_PARAMS_ALL
#define CDX2_PARAMS_ALL 0           // non-pipelined position only: 1 = every wave issues all five parameter loads (see load_params)
#endif
#define CDX2_N_CUS 256                            /* MI355X: 8 XCDs x 32 CUs */
#define CDX2_GETREG_XCC_ID ((3 << 11) | 20)      /* s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4): the XCD this wave runs on, 0-7 */
#ifndef CDX2_XCHG_FAST
#define CDX2_XCHG_FAST 1            // 0: grouped ops exchange through split_exchange (epilogue -> barrier -> publish -> collect, rounds 4-5)
#endif
#ifndef CDX2_XCHG_NOWAIT
#define CDX2_XCHG_NOWAIT 0          // 1 (diagnostic, WRONG results): grouped ops do not collect at all -- what a fully hidden exchange would cost
#endif
#ifndef CDX2_PROF_FETCH
#define CDX2_PROF_FETCH 0           // diagnostic builds: the PROF kernels' stamps 4-6 time the parts of fetch_next (1) / of the exchange (2) instead of the K loop's
#endif
#ifndef CDX2_PIPE_PARAMS
#define CDX2_PIPE_PARAMS 1          // 0: fetch an op's epilogue parameters and the next descriptor at the op's start (round-2 order)
#endif
#ifndef CDX2_MUL24
#define CDX2_MUL24 1                // 0: plain 32-bit multiplies in the per-op address arithmetic (rounds 2-5)
#endif
// Per-lane products of small numbers (LDS offsets, positions x strides: far below 2^23): v_mul_i32_i24 / v_mad_i32_i24 run at full
// rate, v_mul_lo_u32 / v_mad_u64_u32 at a quarter of it -- on the serial per-op path every instruction is ~7 cycles x 40 ops x 20 steps.
static __device__ __forceinline__ int mul24i(int a, int b) { return CDX2_MUL24 ? __mul24(a, b) : a * b; }
static __device__ __forceinline__ int wave_of(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
struct WStream {
    __amdgpu_buffer_rsrc_t rs;
    int lane16;
    __device__ __forceinline__ WStream(const float* wblob, int lane)
        : rs(__builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 0x7fffffff, 0x00020000)), lane16(lane * 16) {}
    // record `q` of the stream that starts at float offset `woff`
    __device__ __forceinline__ f32x4 load(int woff, int q) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, woff * 4 + q * 1024, CDX2_WPOLICY));
    }
};
#pragma clang diagnostic pop

// ---- weight ring -------------------------------------------------------------------------------------------------------
// PF 1-KiB records in flight per wave (a single wave per SIMD has to cover the whole L2 latency by itself: 16; two share it: 8).
template <int PF> struct Ring { f32x4 rec[PF]; };       // head of this wave's next weight stream, issued one op ahead

// K loop of one conv op for this wave: items wave, wave + 4, ...  The first item (and its ring contents) arrive from the
// previous op; further items (layers with more than four tiles) are read from the descriptor / tail table here.
//
// Operand addressing.  Column m of the tile (output position n = m * ostride + ooff) reads input row m * cstride - pad + tap of
// the item's source slot; the slot's HALO2 zero rows make that row valid for every tap, i.e. the B address is LINEAR in
// (tap, chunk): one add per chunk (`+ KSTEP`) and, every `ccn` chunks, one per-lane add for the tap step.  Columns past l_cols
// sit on halo row 0 with a zero tap step.  Nothing else happens per record: wait, 4 MFMAs per (trajectory, column tile), one
// ds_read per (trajectory, column tile), one global load.
template <class M, int NT, int T, int NWV, bool PROF, bool GRP = false>
__device__ __forceinline__ void conv_kloop(const Geom& g, const float* __restrict__ wblob, int vd, const cint* ops,
                                           Item it, int n_items, float* __restrict__ lds, int tf, int lane, int wave,
                                           Ring<WG<NWV>::PF>& ring, unsigned long long* prof, int tune) {
    constexpr int PF = WG<NWV>::PF, NW = NWV;
    // operand ring depth: one wave per SIMD has to cover the ds_read latency with its own MFMAs -- 8 chunks when a chunk is only 4
    // short MFMAs (32 cycles), else 4; with two waves per SIMD the other wave helps and registers are half as many
    constexpr int BD = NWV == 4 ? ((M::KSTEP == 4 && NT * T == 1) ? 8 : 4)
                                : (NT * T == 1 ? (M::KSTEP == 4 ? 8 : 4) : (NT * T == 2 ? 4 : 2));      // (three trajectories: 2)
    // accumulators per tile: a record's four MFMAs on one tile must not form a dependent chain (a 4x4x1 MFMA is 2 passes; the
    // compiler pads dependent pairs with s_nop): 4 independent MFMAs between dependent ones is enough
    // (NA must not depend on T: the summation order of a trajectory is the same whether it shares a workgroup or not)
    // (a 16x16x4 MFMA occupies the pipe as long as its result takes: two accumulators are plenty)
    constexpr int NA = NWV == 4 ? 4 : ((M::KSTEP == 4 && NT == 1) ? 4 : 2);
    static_assert(PF % BD == 0, "operand ring must divide the weight ring");
    const int ptid = (wave == 0 && lane == 0) ? 0 : 1;          // stamp() fires for tid == 0 only
    for (int item = wave; item < n_items; item += NW) {
        if (item != wave) it = load_item(ops + CDX2_DW(vd, CDX2_W2_ITEMS) + (item - NW) * CDX2_ITEM_WORDS);
        const int nq = it.nq, ccn = it.ccn;
        int cc = it.cc;
        int cur[NT], tstep[NT], mcol[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            mcol[nt] = it.col0 + nt * M::COLS + M::col(lane);
            const bool valid = mcol[nt] < g.l_cols;
            // (per-lane products of small numbers: v_mul_i32_i24 is full rate, v_mul_lo_u32 a quarter of it)
            int mrow = mul24i(mcol[nt], g.cstride);
            if (GRP && g.grows) mrow = mul24i(mcol[nt] >> g.gsh, g.grows) + mul24i(mcol[nt] & ((1 << g.gsh) - 1), g.cstride);
            const int row = valid ? mrow - it.pad + CDX2_HALO2 + it.tap : 0;
            cur[nt] = it.src + __mul24(row, it.sstr) + M::koff(lane) + cc * M::KSTEP;
            tstep[nt] = (valid ? it.sstr : 0) - ccn * M::KSTEP;
        }
        if (PROF && !CDX2_PROF_FETCH && prof && item == 0) { asm volatile("" ::"s"(nq), "v"(cur[0])); stamp(prof + 4, ptid); }
        f32x4 acc[T][NT][NA];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < NA; ++j) acc[t][nt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        const WStream ws(wblob, lane);
        const int woff = it.woff;
#define CDX2_WLOAD(q) ws.load(woff, (q))
        f32x4 wr[PF];
        if (item == wave) {                                    // head of the stream was issued during the previous op
#pragma unroll
            for (int u = 0; u < PF; ++u) wr[u] = ring.rec[u];
        } else {
#pragma unroll
            for (int u = 0; u < PF; ++u) wr[u] = CDX2_WLOAD(u);   // unconditional (the blob is padded by PF records)
        }
        // B-operand ring: with ONE wave per SIMD nothing else hides the ~100+ cycle ds_read latency, so the operand of chunk
        // q + BD - 1 is requested before chunk q's MFMAs issue (a chunk is only 32 cycles of matrix pipe in the 4x4 mode).
        f32x4 bq[BD][T][NT];
        auto fetch = [&](f32x4 (&bv)[T][NT]) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[t][nt] = *reinterpret_cast<const f32x4*>(lds + t * tf + cur[nt]);
        };
        auto advance = [&]() {                                 // cursor -> next chunk
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) cur[nt] += M::KSTEP;
            if (++cc == ccn) {                                 // ... -> next tap: one more add
                cc = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) cur[nt] += tstep[nt];
            }
        };
        // steady-state eligibility (see below): 4x4 layers whose taps span a multiple of PF chunks, K slice starting on such a
        // boundary (the host cuts long streams that way), at least one full revolution before the drain
        // (GRP kernels: also the 16x16 streams -- a grouped op is 40 records of 128 matrix-pipe cycles per wave, and the general loop's
        //  conditional refills make the compiler wait for the whole vector-memory queue per record: measured 375 cycles per record)
        const bool aligned = (M::KSTEP == 4 || GRP) && (ccn % PF) == 0 && (it.cc % PF) == 0 && nq >= 2 * PF;
        if (aligned) {                                         // the first BD - 1 chunks lie inside one tap: immediate offsets
#pragma unroll
            for (int j = 0; j < BD - 1; ++j)
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        bq[j][t][nt] = *reinterpret_cast<const f32x4*>(lds + t * tf + cur[nt] + j * M::KSTEP);
        } else {
#pragma unroll
            for (int j = 0; j < BD - 1; ++j) {
                if (j < nq) {
                    if (j > 0) advance();
                    fetch(bq[j]);
                }
            }
        }
        if (PROF && !CDX2_PROF_FETCH && prof && item == 0) { asm volatile("" ::"v"(wr[0][0]), "v"(bq[0][0][0][0])); stamp(prof + 5, ptid); }
        // two waves per SIMD: the arbiter favours the older wave (0-3), which then finishes its K loop well before its
        // SIMD-mate and leaves it running alone at the single-wave rate; raising the younger wave's priority evens them out

        // chunk at ring slot u (record index == u mod PF, PF % BD == 0 -> operand slot u % BD is static after unrolling)
        // (the operand fetch is UNconditional -- past the item's last chunk it re-reads the last valid address -- and the loop below
        //  leaves through one exit: with a fetch on only some paths the compiler cannot count the LDS queue any more and puts an
        //  lgkmcnt(0) in front of every chunk's MFMAs, i.e. the full ds_read latency per record: measured 255 instead of ~130
        //  cycles per 16x16 record)
        auto chunk = [&](const f32x4 a, const int u, bool more) {
            if (more) advance();
            fetch(bq[(u + BD - 1) % BD]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt][j % NA] = M::mfma(a[j], bq[u % BD][t][nt][j], acc[t][nt][j % NA]);
        };

        // The refill of a record is issued one chunk LATE (after the next chunk's MFMAs): the scheduler may hoist a load
        // over the MFMAs of its own basic block, and a refill overlapping the last reads of the value it replaces makes the
        // register allocator double-buffer the whole ring (16 v_mov_b64 + a vmcnt(0) drain per revolution, seen in the ISA).
        // Lagged by a chunk, the old value is dead a full basic block earlier and every slot keeps its registers.
        auto refill = [&](const int slot, int q) { wr[slot] = CDX2_WLOAD(q); };
        // steady state (4x4 layers whose taps span a multiple of PF chunks -- C_in >= 64 -- with the K slice starting on such a
        // boundary, which the host guarantees): a revolution of PF chunks then lies inside ONE tap, so every operand address of
        // the revolution is `base + constant` -- the ds_read carries the chunk as an immediate offset and nothing but
        // {wait, MFMAs, ds_read, global_load} is issued per record; the tap step is applied once per revolution.  With one wave
        // per SIMD the instruction count IS the speed of this loop (measured: 17 instructions per record = 100 cycles for 32
        // cycles of matrix work).  16x16 layers are short (tens of records) and take the general loop below.
        int qi = 0;
        if (aligned) {
            const int n_main = (nq / PF - 1) * PF;             // >= PF
            // operand base of the current revolution's first chunk (bytes from the trajectory region), and of the next one's
            int rb[NT], rbn[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rb[nt] = cur[nt] * 4;
            int cc0 = it.cc;                                   // chunk-in-tap of the revolution's first chunk
            const char* ldsb = reinterpret_cast<const char*>(lds);
            for (; qi < n_main; qi += PF) {
                // two waves per SIMD: the arbiter favours the older wave (0-3), which then leaves a long K loop thousands of cycles before
                // its SIMD-mate and lets it finish alone at the single-wave rate.  Alternating the priority every revolution keeps
                // the pair level (+0.5-1 %; raising the younger waves for the whole loop only moves the skew to the other side).
                if (NWV == 8) {
                    if (((qi / PF) ^ (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                const bool wrap = cc0 + PF == ccn;             // the NEXT revolution starts the next tap
                cc0 = wrap ? 0 : cc0 + PF;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) rbn[nt] = rb[nt] + PF * M::KSTEP * 4 + (wrap ? tstep[nt] * 4 : 0);
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int ua = u + BD - 1;                 // chunk whose operand is requested now
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            bq[ua % BD][t][nt] = *reinterpret_cast<const f32x4*>(
                                ldsb + (ua < PF ? rb[nt] : rbn[nt]) + t * tf * 4 + (ua % PF) * M::KSTEP * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < T; ++t)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[t][nt][j % NA] = M::mfma(wr[u][j], bq[u % BD][t][nt][j], acc[t][nt][j % NA]);
                    if (u > 0) refill(u - 1, qi + u - 1 + PF);
                    else if (qi > 0) refill(PF - 1, qi - 1 + PF);
                    // one refill per chunk, in place: left to itself the scheduler batches the 16 loads of a revolution into
                    // 2-3 bursts and the ring spends half of the time 6-9 deep instead of 16 (seen in the ISA and the stream rate)
                    // (with ONE trajectory per workgroup the scheduler's own placement is 1.2 % faster; with two, the fence is 1.3 %
                    //  faster -- A/B on MI355X, tools/gpu_variants.sh)
                    if (NWV == 4 || T >= 2) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) rb[nt] = rbn[nt];
            }
            // hand the cursor to the general loop: it sits on chunk qi + BD - 2
            cc = cc0 + BD - 2;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) cur[nt] = rb[nt] / 4 + (BD - 2) * M::KSTEP;
        }
        // drain: the last (up to 2*PF - 1) records, refilling only while records remain
        for (; qi < nq; qi += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (qi + u >= nq) break;                       // single exit: every chunk has exactly one predecessor
                chunk(wr[u], u, qi + u + BD - 1 < nq);
                const int q = qi + u - 1 + PF;                 // lagged refill of the previous position
                if (q >= PF && q < nq) refill((u + PF - 1) % PF, q);
            }
        }
        if (NWV == 8) __builtin_amdgcn_s_setprio(0);
        if (PROF && !CDX2_PROF_FETCH && prof && item == 0) { asm volatile("" ::"v"(acc[0][0][0][0]), "v"(acc[0][0][1][0])); stamp(prof + 6, ptid); }
        // D fragment: 4 consecutive rows (channels) of one column -> stage[k slice][output position][row tile + rows]
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (mcol[nt] < g.l_cols) {
                    const int n = mul24i(mcol[nt], g.ostride) + it.ooff;
                    f32x4 dd = acc[t][nt][0];
                    if (NA == 2) dd += acc[t][nt][NA - 1];
                    if (NA == 4) dd = (acc[t][nt][0] + acc[t][nt][1]) + (acc[t][nt][NA / 2] + acc[t][nt][NA - 1]);
                    *reinterpret_cast<f32x4*>(lds + t * tf + g.stage + it.part + mul24i(n, g.sstride) + M::drow(lane)) = dd;
                }
            }
    }
}

// The epilogue's view of a descriptor (decoded with v_readlane right where it is needed: SGPR live ranges stay short).
struct EpiDesc {
    int flags, c_out, l_out, coutp, sstride, ksplit, dst, dstride, res, rstride, shift, nk;
    int save, savestr, stats, dst2, d2stride;          // backward-pass extras (F2_SAVE / F2_GNBWD / F2_DUAL)
    int kpost;                                         // partial tiles (after the first ksplit) that are added AFTER norm / activation
    float inv_cnt;
    float odiv;                                        // F2_OUT_DIV divisor (forward ops: the word W2_SAVE_STRIDE)
    int cgreal4;                                       // F2_COLNORM: float4 items per lane group holding real channels, 0 = all
};
template <bool BWD>
__device__ __forceinline__ EpiDesc decode_epi(int vd) {
    EpiDesc e;
    e.flags = CDX2_DW(vd, CDX2_W2_FLAGS); e.c_out = CDX2_DW(vd, CDX2_W2_COUT); e.l_out = CDX2_DW(vd, CDX2_W2_LOUT);
    e.coutp = CDX2_DW(vd, CDX2_W2_COUTP); e.sstride = CDX2_DW(vd, CDX2_W2_SSTRIDE); e.ksplit = CDX2_DW(vd, CDX2_W2_KSPLIT);
    e.dst = CDX2_DW(vd, CDX2_W2_DST); e.dstride = CDX2_DW(vd, CDX2_W2_DST_STRIDE); e.res = CDX2_DW(vd, CDX2_W2_RES);
    e.rstride = CDX2_DW(vd, CDX2_W2_RES_STRIDE); e.shift = CDX2_DW(vd, CDX2_W2_CG4_SHIFT); e.nk = CDX2_DW(vd, CDX2_W2_NK);
    e.inv_cnt = __int_as_float(CDX2_DW(vd, CDX2_W2_INV_CNT));
    e.kpost = CDX2_DW(vd, CDX2_W2_KPOST);
    e.odiv = BWD ? 1.0f : __int_as_float(CDX2_DW(vd, CDX2_W2_ODIV));
    e.cgreal4 = BWD ? 0 : CDX2_DW(vd, CDX2_W2_CGREAL4);
    e.save = e.savestr = e.stats = e.dst2 = e.d2stride = 0;
    if (BWD && (e.flags & (CDX2_F2_SAVE | CDX2_F2_GNBWD))) {
        e.save = CDX2_DW(vd, CDX2_W2_SAVE); e.savestr = CDX2_DW(vd, CDX2_W2_SAVE_STRIDE); e.stats = CDX2_DW(vd, CDX2_W2_STATS);
        e.dst2 = CDX2_DW(vd, CDX2_W2_DST2); e.d2stride = CDX2_DW(vd, CDX2_W2_DST2_STRIDE);
    }
    return e;
}

struct EpiParams { f32x4 bi, ga, be, em, pb, sc; };  // bias, gamma, beta, FiLM vector, bias of the post-norm extra conv, FiLM scale (F2_FILM, COND kernels)

// two half-wave sums at once (the DPP chains of a and b interleave, so the second one is almost free)
__device__ __forceinline__ void half_sum2(float& a, float& b, int lane) {
    a = dpp_add<0xB1>(a);  b = dpp_add<0xB1>(b);
    a = dpp_add<0x4E>(a);  b = dpp_add<0x4E>(b);
    a = dpp_add<0x141>(a); b = dpp_add<0x141>(b);
    a = dpp_add<0x140>(a); b = dpp_add<0x140>(b);
    const int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b);
    const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 0)), a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 16));
    const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 32)), a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 48));
    const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 0)), b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 16));
    const float b2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 32)), b3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 48));
    a = lane < 32 ? a0 + a1 : a2 + a3;
    b = lane < 32 ? b0 + b1 : b2 + b3;
}

// Epilogue of one op for one trajectory region `tl`.  Thread -> (group g = tid / 32, float4 item li + 32 k): 4 consecutive
// channels c..c+3 of position pos0 + k * pstep.  NK = items per lane (compile-time so the values stay in registers).
// GroupNorm statistics in ONE cross-lane round: sums of (x - s) and (x - s)^2 with s = the group's first element (no E[x^2] -
// E[x]^2 cancellation; the second dependent reduction of a two-pass scheme is ~150 cycles of pure latency per op).
// Grouped ops, fast exchange: the epilogue thread that produced an item also publishes it -- straight from its registers into the
// group's tile in L2 (granule format of split_exchange), tile position `vbase + pos`.
struct Pub { float* tile; float tag; int vbase, coutp; bool on; };
template <int NK, bool BWD, bool COND = false, bool MLP = false, bool PUBLISH = false>
__device__ __forceinline__ void epilogue(float* __restrict__ tl, const EpiParams& P, const EpiDesc& e, int stage, int c, int pos0,
                                         int pstep, int li, int nv, int lane, int grp, float* __restrict__ ws, const Pub* pub = nullptr) {
    f32x4 v[NK];
    bool ok[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        ok[k] = li + 32 * k < nv;
        const int pos = ok[k] ? pos0 + k * pstep : 0;
        // bias + partial 0 + partial 1 + ... in this order; the reads go out four / two at a time so that their ~130-cycle
        // latencies overlap instead of adding up (8 waves: up to 8 K slices per tile)
        f32x4 acc = P.bi;
        const float* sp = tl + stage + mul24i(pos, e.sstride) + c;
        const int kstep = e.l_out * e.sstride;
        int ks = 0;
        for (; ks + 4 <= e.ksplit; ks += 4, sp += 4 * kstep) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sp), p1 = *reinterpret_cast<const f32x4*>(sp + kstep);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(sp + 2 * kstep), p3 = *reinterpret_cast<const f32x4*>(sp + 3 * kstep);
            acc += p0; acc += p1; acc += p2; acc += p3;
        }
        if (ks + 2 <= e.ksplit) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sp), p1 = *reinterpret_cast<const f32x4*>(sp + kstep);
            acc += p0; acc += p1;
            ks += 2; sp += 2 * kstep;
        }
        if (ks < e.ksplit) acc += *reinterpret_cast<const f32x4*>(sp);
        v[k] = acc;
    }
    const int act_id = MLP ? (e.flags >> CDX2_F2_ACT_SHIFT) & 15 : 0;
    if (MLP && (e.flags & CDX2_F2_COLNORM)) {
        // per-sample GroupNorm (reference pearcemlp.py FCBlock): the statistics of ONE position over the group's channels = the
        // 2^shift float4 items of that position, which sit in adjacent lanes; two passes (mean, centred squares)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float s1 = ok[k] ? (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]) : 0.f;
            s1 = seg_sum(s1, e.shift, lane);
            const float mean = s1 * e.inv_cnt;
            const f32x4 dl = v[k] - mean;
            // (groups narrower than the lane group: the pad channels hold exact zeros and must not count as (0 - mean)^2)
            const bool realc = e.cgreal4 == 0 || (li & ((1 << e.shift) - 1)) < e.cgreal4;
            float s2 = (ok[k] && realc) ? (dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]) : 0.f;
            s2 = seg_sum(s2, e.shift, lane);
            const float rstd = __builtin_amdgcn_rsqf(s2 * e.inv_cnt + CDX_GN_EPS);
            const f32x4 y = dl * rstd * P.ga + P.be;
            v[k] = act_id ? (f32x4){act2_f(y[0], act_id), act2_f(y[1], act_id), act2_f(y[2], act_id), act2_f(y[3], act_id)}
                          : (f32x4){mish2(y[0]), mish2(y[1]), mish2(y[2]), mish2(y[3])};
        }
    } else if (e.flags & CDX2_F2_GN) {
        const int i0 = __builtin_bit_cast(int, v[0][0]);
        const float f0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i0, 0));
        const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i0, 32));
        const float sh = lane < 32 ? f0 : f1;                  // first element of this half-wave's group
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const f32x4 dl = v[k] - sh;
            s1 += ok[k] ? (dl[0] + dl[1]) + (dl[2] + dl[3]) : 0.f;
            s2 += ok[k] ? (dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]) : 0.f;
        }
        half_sum2(s1, s2, lane);
        const float m1 = s1 * e.inv_cnt;
        const float mean = sh + m1;
        const float rstd = __builtin_amdgcn_rsqf(s2 * e.inv_cnt - m1 * m1 + CDX_GN_EPS);
        const bool keep = BWD && (e.flags & CDX2_F2_SAVE) != 0;  // the backward pass of this layer wants x_hat and rstd
        if (keep && li == 0) tl[e.stats + grp] = rstd;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const f32x4 xh = (v[k] - mean) * rstd;
            if (keep && ok[k] && c < e.c_out) {
                // (F2_SAVE_GLOBAL: the trajectory's block of the launch workspace instead of LDS -- two-trajectory guided programs)
                float* sv = ((e.flags & CDX2_F2_SAVE_GLOBAL) ? ws : tl) + e.save + mul24i(pos0 + k * pstep, e.savestr) + c;
                *reinterpret_cast<f32x4*>(sv) = xh;
            }
            const f32x4 y = xh * P.ga + P.be;
            v[k] = (MLP && act_id) ? (f32x4){act2_f(y[0], act_id), act2_f(y[1], act_id), act2_f(y[2], act_id), act2_f(y[3], act_id)}
                                   : (f32x4){mish2(y[0]), mish2(y[1]), mish2(y[2]), mish2(y[3])};
        }
    } else if (MLP && act_id > 1) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
            v[k] = (f32x4){act2_f(v[k][0], act_id), act2_f(v[k][1], act_id), act2_f(v[k][2], act_id), act2_f(v[k][3], act_id)};
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (!ok[k]) continue;
        const int pos = pos0 + k * pstep;
        f32x4 y = v[k];
        if (COND && (e.flags & CDX2_F2_FILM)) {
            // ChiUNet1d's FiLM (reference chiunet.py:41-45: scale * h + bias, two ATen ops): two roundings, no fma contraction
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = __fadd_rn(__fmul_rn(P.sc[j], y[j]), P.em[j]);
        } else if (e.flags & CDX2_F2_EMB) y += P.em;
        if (e.kpost) {
            // the ResidualBlock's 1x1 skip conv (reference jannerunet.py:58, :69) was computed by other waves of this op: bias + its
            // partial tiles, added after the norm / activation
            f32x4 pv = P.pb;
            const float* pp = tl + stage + mul24i(e.ksplit * e.l_out + pos, e.sstride) + c;
            for (int j = 0; j < e.kpost; ++j) pv += *reinterpret_cast<const f32x4*>(pp + j * e.l_out * e.sstride);
            y += pv;
        }
        if (e.flags & CDX2_F2_RES) y += *reinterpret_cast<const f32x4*>(tl + e.res + mul24i(pos + CDX2_HALO2, e.rstride) + c);
        if (MLP && (e.flags & CDX2_F2_OUT_DIV)) {             // PearceMlp's h / 1.414 (reference pearcemlp.py:64), a true division
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = __fdiv_rn(y[j], e.odiv);
        }
        if (PUBLISH && pub->on) {
            f32x4* gq = reinterpret_cast<f32x4*>(pub->tile + (size_t)(mul24i(pub->vbase + pos, pub->coutp) + c) * 2);
            gq[0] = (f32x4){y[0], pub->tag, y[1], pub->tag};
            gq[1] = (f32x4){y[2], pub->tag, y[3], pub->tag};
        }
        float* o = tl + e.dst + mul24i(pos + CDX2_HALO2, e.dstride) + c;
        if (c + 3 < e.c_out) {
            *reinterpret_cast<f32x4*>(o) = y;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < e.c_out) o[j] = y[j];
        }
    }
}

// d Mish(a) / d a with tanh(softplus(a)) = n / (n + 2), n = e^a (e^a + 2):  n/(n+2) + a * 4 e^a (e^a + 1) / (n + 2)^2
__device__ __forceinline__ float mish2_grad(float a) {
    const float e = __expf(fminf(a, 20.0f));
    const float n = e * (e + 2.0f);
    const float r = __builtin_amdgcn_rcpf(n + 2.0f);
    return n * r + a * (4.0f * e * (e + 1.0f)) * (r * r);
}

// Backward epilogue (F2_GNBWD): v = staged conv result [+ residual slot]; [dst2 <- v]; then the backward of y = Mish(gamma x_hat +
// beta), x_hat = (u - mean) rstd of the layer whose x_hat / rstd the forward pass saved:
//   g_xhat = v * Mish'(gamma x_hat + beta) * gamma;   dst <- rstd (g_xhat - mean_grp(g_xhat) - x_hat mean_grp(g_xhat x_hat))
// (torch's native_group_norm_backward for the input, reference utils/building_blocks.py:60-76 + nn.Mish under autograd).
template <int NK>
__device__ __forceinline__ void epilogue_bwd(float* __restrict__ tl, const EpiParams& P, const EpiDesc& e, int stage, int c, int pos0,
                                             int pstep, int li, int nv, int lane, int grp, const float* __restrict__ ws) {
    f32x4 xh_pre[NK];
    const float* __restrict__ svb = ((e.flags & CDX2_F2_SAVE_GLOBAL) ? ws : tl) + e.save;
#pragma unroll
    for (int k = 0; k < NK; ++k) {      // saved x_hat first: from the global workspace this is the longest latency of the epilogue
        const bool okk = li + 32 * k < nv;
        const int pos = okk ? pos0 + k * pstep : 0;
        xh_pre[k] = c < e.c_out ? *reinterpret_cast<const f32x4*>(svb + mul24i(pos, e.savestr) + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 gx[NK], xh[NK];
    bool ok[NK];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        ok[k] = li + 32 * k < nv;
        const int pos = ok[k] ? pos0 + k * pstep : 0;
        f32x4 acc = P.bi;
        const float* sp = tl + stage + mul24i(pos, e.sstride) + c;
        const int kstep = e.l_out * e.sstride;
        int ks = 0;
        for (; ks + 4 <= e.ksplit; ks += 4, sp += 4 * kstep) {      // same 4 / 2 / 1 cascade as the forward epilogue
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sp), p1 = *reinterpret_cast<const f32x4*>(sp + kstep);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(sp + 2 * kstep), p3 = *reinterpret_cast<const f32x4*>(sp + 3 * kstep);
            acc += p0; acc += p1; acc += p2; acc += p3;
        }
        if (ks + 2 <= e.ksplit) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sp), p1 = *reinterpret_cast<const f32x4*>(sp + kstep);
            acc += p0; acc += p1;
            ks += 2; sp += 2 * kstep;
        }
        if (ks < e.ksplit) acc += *reinterpret_cast<const f32x4*>(sp);
        if (e.flags & CDX2_F2_RES) acc += *reinterpret_cast<const f32x4*>(tl + e.res + mul24i(pos + CDX2_HALO2, e.rstride) + c);
        if ((e.flags & CDX2_F2_DUAL) && ok[k]) *reinterpret_cast<f32x4*>(tl + e.dst2 + mul24i(pos + CDX2_HALO2, e.d2stride) + c) = acc;
        // (lane groups past C_out -- nets with fewer than 8 x 4 channels -- have nothing saved: zeros, never stored)
        xh[k] = xh_pre[k];
        const f32x4 a = xh[k] * P.ga + P.be;
        const f32x4 d = (f32x4){mish2_grad(a[0]), mish2_grad(a[1]), mish2_grad(a[2]), mish2_grad(a[3])};
        gx[k] = acc * d * P.ga;
        const f32x4 gh = gx[k] * xh[k];
        s1 += ok[k] ? (gx[k][0] + gx[k][1]) + (gx[k][2] + gx[k][3]) : 0.f;
        s2 += ok[k] ? (gh[0] + gh[1]) + (gh[2] + gh[3]) : 0.f;
    }
    half_sum2(s1, s2, lane);
    const float m1 = s1 * e.inv_cnt, m2 = s2 * e.inv_cnt;
    const float rstd = tl[e.stats + grp];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (!ok[k]) continue;
        const int pos = pos0 + k * pstep;
        const f32x4 gu = (gx[k] - m1 - xh[k] * m2) * rstd;
        float* o = tl + e.dst + mul24i(pos + CDX2_HALO2, e.dstride) + c;
        if (c + 3 < e.c_out) {
            *reinterpret_cast<f32x4*>(o) = gu;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < e.c_out) o[j] = gu[j];
        }
    }
}

// Classifier head, forward and backward in one op (KIND2_HEAD; reference nn_classifier/half_jannerunet.py:49-50, :62):
// z_j = e_j + sum_{l,c} W1[l][c][j] x[l][c];  gz_j = w2_j Mish'(z_j);  dst[l][c] = sum_j W1[l][c][j] gz_j.
template <int THREADS>
__device__ __forceinline__ void run_head(const cdx_unet2_launch& L, int vd, const float* __restrict__ emb_row, float* __restrict__ tl,
                                         int tid) {
    const int hidden = CDX2_DW(vd, CDX2_W2_COUT), len = CDX2_DW(vd, CDX2_W2_LOUT), ch = CDX2_DW(vd, CDX2_W2_LCOLS);
    const int src = CDX2_DW(vd, CDX2_W2_RES), sstr = CDX2_DW(vd, CDX2_W2_RES_STRIDE);
    const int dst = CDX2_DW(vd, CDX2_W2_DST), dstr = CDX2_DW(vd, CDX2_W2_DST_STRIDE);
    const float* __restrict__ w1 = L.wblob + CDX2_DW(vd, CDX2_W2_BOFF);
    const float* __restrict__ w2 = L.wblob + CDX2_DW(vd, CDX2_W2_GAMMA);
    const float* __restrict__ ev = emb_row + CDX2_DW(vd, CDX2_W2_EMB);
    float* gz = tl + L.stage_off;
    for (int j = tid; j < hidden; j += THREADS) {
        float z = ev[j];
        for (int l = 0; l < len; ++l) {
#pragma unroll 8
            for (int c = 0; c < ch; ++c) z = fmaf(w1[(size_t)(l * ch + c) * hidden + j], tl[src + (l + CDX2_HALO2) * sstr + c], z);
        }
        gz[j] = w2[j] * mish2_grad(z);
    }
    __syncthreads();
    for (int i = tid; i < len * ch; i += THREADS) {
        const int l = i / ch, c = i - l * ch;
        const float* wr = w1 + (size_t)i * hidden;
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < hidden; ++j) acc = fmaf(wr[j], gz[j], acc);
        tl[dst + (l + CDX2_HALO2) * dstr + c] = acc;
    }
    for (int i = tid; i < 2 * CDX2_HALO2 * dstr; i += THREADS) {       // halo rows of the gradient slot (a conv source)
        const int r = i / dstr, col = i - r * dstr;
        tl[dst + (r < CDX2_HALO2 ? r : len + r) * dstr + col] = 0.f;
    }
    __syncthreads();
}

// Classifier head, forward only: y = b2 + sum_j w2_j Mish(z_j) -> out (the final log_p of a guided launch).  [w2 | b2] lie together.
template <int THREADS>
__device__ __forceinline__ void run_head_fwd(const cdx_unet2_launch& L, int vd, const float* __restrict__ emb_row, float* __restrict__ tl,
                                             int tid, float* __restrict__ out) {
    const int hidden = CDX2_DW(vd, CDX2_W2_COUT), len = CDX2_DW(vd, CDX2_W2_LOUT), ch = CDX2_DW(vd, CDX2_W2_LCOLS);
    const int src = CDX2_DW(vd, CDX2_W2_RES), sstr = CDX2_DW(vd, CDX2_W2_RES_STRIDE);
    const float* __restrict__ w1 = L.wblob + CDX2_DW(vd, CDX2_W2_BOFF);
    const float* __restrict__ w2 = L.wblob + CDX2_DW(vd, CDX2_W2_GAMMA);
    const float* __restrict__ ev = emb_row + CDX2_DW(vd, CDX2_W2_EMB);
    float* hz = tl + L.stage_off;
    for (int j = tid; j < hidden; j += THREADS) {
        float z = ev[j];
        for (int l = 0; l < len; ++l) {
#pragma unroll 8
            for (int c = 0; c < ch; ++c) z = fmaf(w1[(size_t)(l * ch + c) * hidden + j], tl[src + (l + CDX2_HALO2) * sstr + c], z);
        }
        hz[j] = w2[j] * mish2(z);
    }
    __syncthreads();
    if (tid == 0 && out != nullptr) {
        float y = w2[hidden];
        for (int j = 0; j < hidden; ++j) y += hz[j];
        *out = y;
    }
    __syncthreads();
}

// Issue the first PF records of item `it` (this wave's first item of the next op) into the ring.  No clamp to the item's
// record count: the blob ends with PF records of padding, slots past `nq` are simply never consumed.
template <int PF>
__device__ __forceinline__ void prefetch_ring(const Item& it, const float* __restrict__ wblob, int lane, Ring<PF>& ring) {
    const WStream ws(wblob, lane);
#pragma unroll
    for (int u = 0; u < PF; ++u) ring.rec[u] = ws.load(it.woff, u);
}

// The five per-channel epilogue parameters of the op whose descriptor view is `vd` (bias, post-norm bias, gamma, beta, FiLM vector).
// ALL = false: only the waves that run epilogues load, and only what the op's flags ask for (a 64-lane float4 load is 1 KiB through the
// CU's address path, ~16 cycles each: 8 waves x 5 unconditional loads cost ~450 cycles per op, measured).  That is the right form when
// the loads sit after the K loop (PIPE).  ALL = true: every wave issues all five (an unused one re-reads the bias) -- for the round-2
// position at the op's start, where loads on only some paths make hipcc fall back to `s_waitcnt vmcnt(0)` in the K loop.
template <bool COND, bool SPLIT_T, bool ALL, bool MLP = false, bool GRP = false>
__device__ __forceinline__ EpiParams load_params(const cdx_unet2_launch& L, int vd, const float* __restrict__ emb_row, int emb_tstride,
                                                 int tid, int wave, bool epi_wave) {
    const int etid = tid & 255, li = etid & 31;
    int grp = etid >> 5;
    if (GRP) {       // grouped op: the half-waves are (trajectory, lane group of this member) pairs -- the channel comes from the lane group
        const int xg = CDX2_DW(vd, CDX2_W2_XG);
        if (xg & CDX2_XG_GOP) grp = (xg & 255) + (grp & ((((xg >> 8) & 255) - (xg & 255)) - 1));
    }
    const int flags = CDX2_DW(vd, CDX2_W2_FLAGS), coutp = CDX2_DW(vd, CDX2_W2_COUTP), shift = CDX2_DW(vd, CDX2_W2_CG4_SHIFT);
    const int c = mul24i(grp, coutp >> 3) + 4 * (li & ((1 << shift) - 1));
    // (batch-tiled MLP programs: a layer whose input has a time-dependent part reads its bias from the step's table row)
    const float* __restrict__ pbi = ((MLP && (flags & CDX2_F2_BIAS_EMB)) ? emb_row : L.wblob) + CDX2_DW(vd, CDX2_W2_BOFF) + c;
    const bool gn = (flags & (CDX2_F2_GN | CDX2_F2_GNBWD)) != 0;
    const float* __restrict__ pem = emb_row + (COND && SPLIT_T ? (wave >> 2) * emb_tstride : 0) + CDX2_DW(vd, CDX2_W2_EMB) + c;
    EpiParams P;
    if (ALL) {
        P.bi = *reinterpret_cast<const f32x4*>(pbi);
        P.pb = *reinterpret_cast<const f32x4*>(CDX2_DW(vd, CDX2_W2_KPOST) ? L.wblob + CDX2_DW(vd, CDX2_W2_PBIAS) + c : pbi);
        P.ga = *reinterpret_cast<const f32x4*>(gn ? L.wblob + CDX2_DW(vd, CDX2_W2_GAMMA) + c : pbi);
        P.be = *reinterpret_cast<const f32x4*>(gn ? L.wblob + CDX2_DW(vd, CDX2_W2_BETA) + c : pbi);
        P.em = *reinterpret_cast<const f32x4*>((flags & CDX2_F2_EMB) ? pem + ((COND && (flags & CDX2_F2_FILM)) ? coutp : 0) : pbi);
        P.sc = P.bi;
        if (COND) P.sc = *reinterpret_cast<const f32x4*>((flags & CDX2_F2_FILM) ? pem : pbi);
        return P;
    }
    P.bi = P.ga = P.be = P.em = P.pb = P.sc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (epi_wave) {
        P.bi = *reinterpret_cast<const f32x4*>(pbi);
        if (CDX2_DW(vd, CDX2_W2_KPOST)) P.pb = *reinterpret_cast<const f32x4*>(L.wblob + CDX2_DW(vd, CDX2_W2_PBIAS) + c);
        if (gn) {
            P.ga = *reinterpret_cast<const f32x4*>(L.wblob + CDX2_DW(vd, CDX2_W2_GAMMA) + c);
            P.be = *reinterpret_cast<const f32x4*>(L.wblob + CDX2_DW(vd, CDX2_W2_BETA) + c);
        }
        if (COND && (flags & CDX2_F2_FILM)) {          // table row: [scale | bias], W2_COUTP floats apart
            P.sc = *reinterpret_cast<const f32x4*>(pem);
            P.em = *reinterpret_cast<const f32x4*>(pem + coutp);
        } else if (flags & CDX2_F2_EMB) P.em = *reinterpret_cast<const f32x4*>(pem);
    }
    return P;
}

// What an op needs from global memory besides its weight stream, and when it is fetched (PIPE, one and two trajectories per
// workgroup): right after op n's K loop, together with the head of op n+1's weight stream, go out op n+1's epilogue parameters and op
// n+2's descriptor.  They land during op n's barrier + epilogue, so an op starts with NOTHING recent in the vector-memory queue.
// Round 2 fetched the parameters and the next descriptor at the op's start: the K loop's first use of the weight ring then waited
// for them too (the compiler cannot order the waits apart across the item loop's header) -- ~700-900 cycles per op, the "decode"
// column of profiles/r02_op_profile_wg0.txt.  Three trajectories per workgroup and programs with backward ops keep the old order (no registers to spare).
struct OpFetch {
    EpiParams P;           // parameters of the op about to run
    int vdn2;              // descriptor view of the op after the next one
};
// Which instantiations pipeline the fetch: not three trajectories per workgroup and not the programs with backward ops (16-20 more
// live VGPRs: 239 -> 256 at T = 2 guided, measured 2.5 % slower).  Measured on MI355X, same box, against the round-2 order
// (gpurun r3e / r3f): T = 1 with the parameter loads in FRONT of the staging barrier 4.205 vs 4.227 ms (behind it: 4.264);
// T = 2 with the loads BEHIND the barrier 5.604 vs 5.659 ms (in front: 5.72) -- hence PARAMS_AFTER_BARRIER = (T >= 2).
template <int T, bool BWD> constexpr bool pipe_params() { return CDX2_PIPE_PARAMS && T < 3 && !BWD; }

// Split programs (one trajectory over k workgroups of one XCD; engine/program2.py:compile_janner2_split): what a member knows about
// its group.  After an op that is cut over the members (descriptor word W2_XG) every member publishes the channels it computed into
// the group's tile in global memory and collects the others'.  All members sit on the same XCD, i.e. behind the same L2, so no
// agent-scope fence is needed -- and no flag either: the tile is made of 8-byte {value, tag} granules, tag = the exchange's sequence
// number, written with plain 16-byte stores and polled with L1-bypassing 16-byte loads until both tags of a load match (one L2 round
// trip per exchange instead of store-drain + flag + data; tools/xwg_exchange_probe.hip measured the flag form at ~0.8 k cycles on top
// of the local traffic, in this kernel it cost ~5.7 k cycles per op with its four barriers).  The two tiles of a group alternate by
// the parity of the sequence number: a member can only be two exchanges ahead of another after that one has finished reading.
// Every poll is bounded: a granule that never arrives sets `err` and the launch ends with wrong numbers instead of hanging the GPU.
// Groups are FORMED at run time, not assumed: HIP promises no workgroup -> XCD placement ("workgroup i runs on XCD i % 8" is an
// observation, and a launch that deviates from it would exchange through two different L2s, which are not coherent with each other).
// Every split / grouped launch has exactly 256 workgroups of one per CU (the LDS request sees to that), i.e. exactly 32 on every XCD,
// all resident; in its prologue a workgroup reads HW_REG_XCC_ID and draws a ticket from its XCD's counter (an atomic only workgroups
// behind that same L2 ever touch): ticket t on XCD x = member t % k of group x * (32 / k) + t / k.  The members of a group therefore
// share an L2 by construction, whatever the dispatcher did.  The counters only ever count up (launch field xtick0 = 32 x the launches
// so far); which trajectory a workgroup works on follows from its group, not from its blockIdx.
struct XState {
    int m, k;                  // this workgroup's member index, members per trajectory / trajectories per group
    unsigned seq;              // exchanges done so far (the same number in every member: they run the same op list)
    bool dead;                 // this thread lost a granule: no more waiting in this launch
    // (the group's index -- drawn in the kernel's prologue -- lives in the LDS word behind the trajectory region, not here: one more
    //  scalar live across the op loop measured 1 % at B = 256; every exchange reads it back)
    // (the group's tiles, their size and the error word are re-read from the kernarg segment inside every exchange: five scalar
    //  registers less that would be live -- and spilled -- across the whole op loop)
};

// A failed wait: the error words (pinned host memory, 8 ints: [0] != 0 = failed, then what (1 a granule never came, 2 the ticket of
// this workgroup does not fit a group) / which workgroup / sequence number / item / this workgroup's XCC id / its member index / its
// group) -- written on failure only
__device__ __forceinline__ void xchg_report(int what, const XState& X, int seq, int item, int grp) {
    const KArg* S0 = kernarg();
    asm volatile("" : "+s"(S0));
    int* e = S0->xerr;
    e[1] = what; e[2] = (int)blockIdx.x; e[3] = seq; e[4] = item;
    e[5] = (int)(__builtin_amdgcn_s_getreg(CDX2_GETREG_XCC_ID) & 15); e[6] = X.m; e[7] = grp;
    e[0] = 1;
}

// The group's tile for the exchange with sequence number `seq` (two tiles of 2 * xchg_floats floats per group, alternating by parity).
__device__ __forceinline__ float* exchange_tile(const XState& X, unsigned seq, int grp_idx) {
    const KArg* S0 = kernarg();
    asm volatile("" : "+s"(S0));
    const int xf = S0->xchg_floats;
    return S0->xbuf + (size_t)grp_idx * 4 * xf + (size_t)(seq & 1) * 2 * xf;
}

// `xg` = the op's W2_XG word.  Split programs and grouped ops (XG_GOP): a member owns the lane groups [lo, hi) -- of its one trajectory,
// or (grouped) of all k trajectories of the group, whose sub-slots lie `grows` rows apart in the destination slot; tile position =
// trajectory * l_out + position.  XG_TRAJ: the member owns its WHOLE trajectory (sub-slot X.m; `dst` points at it) and collects the
// other members' trajectories.
// (Measured and dropped in round 4, gpurun r4d vs r4e: publishing straight from the epilogue's registers -- no LDS read-back, no barrier
//  in front of the exchange -- with two collect items in flight per thread was 2.5 % SLOWER at B = 256; what an exchange costs, ~3.6 k
//  cycles, is the wait for the slowest member of the group plus one L2 round trip, not the copy.)
template <int THREADS>
__device__ __forceinline__ void split_exchange(XState& X, int grp_idx, int xg, int gmap, float* __restrict__ tl, int dst, int dstride,
                                               int l_out, int c_out, int coutp, int tid, unsigned long long* prof = nullptr) {
    const int g_lo = xg & 255, g_hi = (xg >> 8) & 255;
    const bool grouped = (xg & (CDX2_XG_GOP | CDX2_XG_TRAJ)) != 0, traj = (xg & CDX2_XG_TRAJ) != 0;
    const int gsh = grouped ? (gmap & 255) : 30, grows = grouped ? (gmap >> 8) : 0;
    const int vl = grouped ? l_out * X.k : l_out;
    const int base = traj ? dst - X.m * grows * dstride : dst;
    const int c4sh = 31 - __builtin_clz(coutp >> 2), cgsh = 31 - __builtin_clz(coutp >> 3);      // (pad32 channel counts: powers of two x 32)
    X.seq += 1;
    float* __restrict__ tile = exchange_tile(X, X.seq, grp_idx);
    const float tag = __uint_as_float(X.seq);
    const int n_items = vl << c4sh;
    bool withhold;                                       // test hook (cdx_unet2_launch.fault): a lost granule on purpose
    {
        const KArg* S0 = kernarg();
        asm volatile("" : "+s"(S0));
        withhold = S0->fault != 0 && X.m == S0->fault - 1;
    }
    // publish: this member's part, read back from the destination slot the epilogue just wrote (pad channels travel along)
    for (int i = tid; i < n_items; i += THREADS) {
        const int vpos = i >> c4sh, c = (i - (vpos << c4sh)) * 4, grp = c >> cgsh;
        const int t = vpos >> gsh, pos = vpos - (t << gsh);
        const bool mine = traj ? t == X.m : (grp >= g_lo && grp < g_hi);
        if (mine && !withhold) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tl + base + mul24i(mul24i(t, grows) + pos + CDX2_HALO2, dstride) + c);
            f32x4* o = reinterpret_cast<f32x4*>(tile + (size_t)(mul24i(vpos, coutp) + c) * 2);
            o[0] = (f32x4){v[0], tag, v[1], tag};
            o[1] = (f32x4){v[2], tag, v[3], tag};
        }
    }
    if (CDX2_PROF_FETCH == 2) stamp(prof ? prof + 5 : nullptr, tid);
    // collect: everybody else's part, straight from L2 (nontemporal loads bypass this CU's L1), as soon as their tags say so.
    // (Measured and dropped, gpurun r4k / r4l: requesting a thread's two items together -- even the same loop merely WRITTEN for two
    //  items with one of them disabled -- is 2 % slower at B = 256 than this plain loop; the poll interval, s_sleep 0 / 1 / 4 / 16, changes
    //  nothing: the exchange waits for the slowest member of the group, not for the polls.)
    for (int i = tid; i < n_items; i += THREADS) {
        const int vpos = i >> c4sh, c = (i - (vpos << c4sh)) * 4, grp = c >> cgsh;
        const int t = vpos >> gsh, pos = vpos - (t << gsh);
        const bool mine = traj ? t == X.m : (grp >= g_lo && grp < g_hi);
        if (!mine && c < c_out) {
            const f32x4* src = reinterpret_cast<const f32x4*>(tile + (size_t)(mul24i(vpos, coutp) + c) * 2);
            f32x4 a, b;
            int spins = 0;
            for (;;) {
                a = __builtin_nontemporal_load(src);
                b = __builtin_nontemporal_load(src + 1);
                if (__float_as_uint(a[1]) == X.seq && __float_as_uint(a[3]) == X.seq && __float_as_uint(b[1]) == X.seq &&
                    __float_as_uint(b[3]) == X.seq)
                    break;
                // (a legitimate wait is tens of microseconds; ~10 ms of polling means the partner is not behind this L2.  Once a thread
                //  has given up it stops waiting altogether: the launch must end; its workgroup stores NaN instead of trajectories)
                if (X.dead || ++spins > 200000) {
                    if (!X.dead) xchg_report(1, X, (int)(X.seq), i, grp_idx);
                    X.dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float* o = tl + base + mul24i(mul24i(t, grows) + pos + CDX2_HALO2, dstride) + c;
            const f32x4 v = (f32x4){a[0], a[2], b[0], b[2]};
            if (c + 3 < c_out) *reinterpret_cast<f32x4*>(o) = v;
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < c_out) o[j] = v[j];
            }
        }
    }
}

// Fast exchange of a grouped op (CDX2_XCHG_FAST): the epilogue waves publish their items from registers (epilogue<..., PUBLISH>) while
// the OTHER four waves -- which have nothing to do during an epilogue but halo rows -- collect the other members' parts: thread u of
// the 256 takes tile item (position u >> nc4sh, float4 u & (nc4 - 1)) of each of the k - 1 other members' channel blocks, requests all of
// them at once (up to six 16-byte loads in flight) and polls until every tag matches.  Round-5 form (split_exchange): epilogue ->
// barrier -> all threads publish from LDS (1.3 k cycles) -> all threads collect, a thread's two items one after the other (1.8-2.7 k)
// -> barrier; here the exchange ends one L2 round trip after the slowest member's epilogue.
__device__ __forceinline__ void collect_fast(XState& X, int grp_idx, unsigned seq, const float* tile, int xg, int gmap,
                                             float* tl, int dst, int dstride, int l_out, int coutp, int u) {
    const int g_lo = xg & 255, w = ((xg >> 8) & 255) - g_lo;                 // lane groups per member (a power of two)
    const int cgsh = 31 - __builtin_clz(coutp >> 3);                         // log2 channels per lane group
    const int nc4sh = (31 - __builtin_clz(w)) + cgsh - 2;                    // log2 float4 items per position and member
    const int gsh = gmap & 255, grows = gmap >> 8;
    const int n_blk = (l_out * X.k) << nc4sh;                                // items of one member's block
    const int km1 = X.k - 1;
    for (int j = u; j < n_blk; j += 256) {
        const int vpos = j >> nc4sh, c4 = j & ((1 << nc4sh) - 1);
        const int t = vpos >> gsh, pos = vpos - (t << gsh);
        const int row = mul24i(mul24i(t, grows) + pos + CDX2_HALO2, dstride);
        const int tbase = mul24i(vpos, coutp);
        f32x4 a[3], b[3];
        int cc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int mb = (X.m + 1 + (r < km1 ? r : 0)) & km1;                // (k = 2: one block; the spare slots re-read it, never stored)
            cc[r] = ((mb * w) << cgsh) + 4 * c4;
            const f32x4* src = reinterpret_cast<const f32x4*>(tile + (size_t)(tbase + cc[r]) * 2);
            a[r] = __builtin_nontemporal_load(src);
            b[r] = __builtin_nontemporal_load(src + 1);
        }
        int pending = (1 << km1) - 1, spins = 0;
        while (pending) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (!(pending & (1 << r))) continue;
                if (__float_as_uint(a[r][1]) == seq && __float_as_uint(a[r][3]) == seq && __float_as_uint(b[r][1]) == seq &&
                    __float_as_uint(b[r][3]) == seq) {
                    *reinterpret_cast<f32x4*>(tl + dst + row + cc[r]) = (f32x4){a[r][0], a[r][2], b[r][0], b[r][2]};
                    pending &= ~(1 << r);
                } else {
                    // (a compiler barrier: the tile is written by OTHER workgroups -- without it the re-read below is a redundant load
                    //  of an address nothing in this function stores to, and is folded into the first one)
                    asm volatile("" ::: "memory");
                    const f32x4* src = reinterpret_cast<const f32x4*>(tile + (size_t)(tbase + cc[r]) * 2);
                    a[r] = __builtin_nontemporal_load(src);
                    b[r] = __builtin_nontemporal_load(src + 1);
                }
            }
            if (pending) {
                // (bounded like split_exchange's polls: a thread that has given up once stops waiting altogether)
                if (X.dead || ++spins > 200000) {
                    if (!X.dead) xchg_report(1, X, (int)seq, j, grp_idx);
                    X.dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
}

// One op.  `vd`: this wave's view of the op's descriptor, `it`: this wave's first item, both fetched during the previous op;
// `vdn`: the next op's descriptor, whose load was issued before this call.  Leaves the next op's first item in `it`.
// Epilogue threads: 256 per trajectory (8 GroupNorm groups x 32 lanes).  4 waves: all of them, one trajectory after the other;
// 8 waves, T = 2: waves 0-3 take trajectory 0 while waves 4-7 take trajectory 1; 8 waves, T = 1: waves 0-3 run the epilogue,
// waves 4-7 rewrite the destination's halo rows.
// SPLIT: THIS op is cut over the members / grouped / exchanged (descriptor word W2_XG != 0): the column map, the grouped epilogue, the
// group-slot halo rows and the exchange are compiled in.  MEMBER: the kernel runs member programs (descriptor offset per member, a
// parameter fetch that understands grouped ops -- the NEXT op may be one).
// (Measured and dropped, gpurun r4m: giving the ordinary ops of a member program their own <SPLIT = false, MEMBER = true> instantiation of
//  this function next to the <true, true> one -- two inlined copies in the op loop, 60 instead of 38 scalar spills -- is 2.6 % SLOWER at
//  B = 256 (3.748 vs 3.650 ms) than one copy whose special paths an ordinary op merely branches around.)
template <int T, int NWV, bool BWD, bool PROF, bool COND, bool MLP = false, bool SPLIT = false, bool MEMBER = SPLIT>
__device__ __forceinline__ void run_op(const cdx_unet2_launch& L, const cint* ops, int vd, int vdn, Item& it,
                                       const float* __restrict__ emb_row, int emb_tstride, float* __restrict__ lds, int tid,
                                       Ring<WG<NWV>::PF>& ring, unsigned long long* prof, int b0, OpFetch& F,
                                       const float* __restrict__ emb_next, int emb_next_tstride, int op_next2, int pass = 0,
                                       XState* X = nullptr) {
    constexpr bool SPLIT_T = NWV == 8 && T >= 2;       // waves 0-3 take trajectories 0, 2; waves 4-7 trajectory 1
    constexpr bool PIPE = pipe_params<T, BWD>();
    constexpr bool PARAMS_AFTER_BARRIER = PIPE && T >= 2;
    // everything the NEXT op needs, issued in one go (see OpFetch)
    // `params`: also the next op's epilogue parameters (ops without a K loop; T = 1).  With two trajectories per workgroup a conv op
    // issues those AFTER its staging barrier instead (see pipe_params)
    auto fetch_next = [&](bool params) {
        it = inline_item(vdn);
        if (PROF && CDX2_PROF_FETCH == 1) stamp(prof ? prof + 4 : nullptr, tid);
        if (wave_of(tid) < CDX2_DW(vdn, CDX2_W2_NITEMS)) prefetch_ring(it, L.wblob, tid & 63, ring);
        if (PROF && CDX2_PROF_FETCH == 1) stamp(prof ? prof + 5 : nullptr, tid);
        if (PIPE) {
            if (params)
                F.P = load_params<COND, SPLIT_T, false, MLP, MEMBER>(L, vdn, emb_next, emb_next_tstride, tid, wave_of(tid), NWV == 4 || SPLIT_T || wave_of(tid) < 4);
            if (PROF && CDX2_PROF_FETCH == 1) stamp(prof ? prof + 6 : nullptr, tid);
            F.vdn2 = load_desc<NWV>(L.ops, op_next2 + (MEMBER ? X->m * L.n_ops : 0), tid & 63, wave_of(tid));
        }
    };
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tf = L.traj_floats;
    if (BWD && CDX2_DW(vd, CDX2_W2_KIND) == CDX2_KIND2_HEAD) {       // classifier head: no K loop, its own two barriers
        fetch_next(true);
#pragma unroll 1
        for (int t = 0; t < T; ++t) run_head<WG<NWV>::THREADS>(L, vd, emb_row, lds + t * tf, tid);
        return;
    }
    if (BWD && CDX2_DW(vd, CDX2_W2_KIND) == CDX2_KIND2_LOADX) {      // the classifier's copy of x_t, from global memory
        fetch_next(true);
        const int dst = CDX2_DW(vd, CDX2_W2_DST), dstr = CDX2_DW(vd, CDX2_W2_DST_STRIDE);
        const int len = CDX2_DW(vd, CDX2_W2_LOUT), ch = CDX2_DW(vd, CDX2_W2_COUT);
        const int b_end = L.traj_first + L.traj_count;
        const KArg* S = kernarg();                                   // (read on demand: see the solver step)
        asm volatile("" : "+s"(S));
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const float* __restrict__ xg = S->x_out + (size_t)(b0 + t) * len * ch;
            const bool real = b0 + t < b_end;
            for (int i = tid; i < (len + 2 * CDX2_HALO2) * dstr; i += WG<NWV>::THREADS) {
                const int r = i / dstr, c = i - r * dstr, n = r - CDX2_HALO2;
                lds[t * tf + dst + i] = (real && n >= 0 && n < len && c < ch) ? xg[n * ch + c] : 0.f;
            }
        }
        __syncthreads();
        return;
    }
    if (MLP && CDX2_DW(vd, CDX2_W2_KIND) == CDX2_KIND2_LOADC) {      // context slot <- the samples' condition features (or zeros)
        fetch_next(true);
        if (pass == 2) return;                                       // (single-forward steps after the first: the slot still holds them)
        const int dst = CDX2_DW(vd, CDX2_W2_DST), dstr = CDX2_DW(vd, CDX2_W2_DST_STRIDE);
        const int len = CDX2_DW(vd, CDX2_W2_LOUT), ch = CDX2_DW(vd, CDX2_W2_COUT);
        const int b_end = L.traj_first + L.traj_count;
        const KArg* S = kernarg();
        asm volatile("" : "+s"(S));
        // (the unconditional forward of a classifier-free-guidance pair and a request without a condition see zeros: the reference
        //  substitutes a zero tensor, e.g. pearcemlp.py:59-60)
        const float* __restrict__ cg = pass == 0 ? S->ctx : nullptr;
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const bool real = cg != nullptr && b0 + t < b_end;
            const float* __restrict__ cb = cg + (size_t)(b0 + t) * len * ch;
            for (int i = tid; i < (len + 2 * CDX2_HALO2) * dstr; i += WG<NWV>::THREADS) {
                const int r = i / dstr, c = i - r * dstr, n = r - CDX2_HALO2;
                lds[t * tf + dst + i] = (real && n >= 0 && n < len && c < ch) ? cb[n * ch + c] : 0.f;
            }
        }
        __syncthreads();
        return;
    }
    const int flags = CDX2_DW(vd, CDX2_W2_FLAGS), coutp = CDX2_DW(vd, CDX2_W2_COUTP), shift = CDX2_DW(vd, CDX2_W2_CG4_SHIFT);
    const int l_out = CDX2_DW(vd, CDX2_W2_LOUT), sstride = CDX2_DW(vd, CDX2_W2_SSTRIDE);
    // grouped op (SPLIT kernels, word W2_XG: XG_GOP): columns = (trajectory of the group) x position.  Only the K loop's column map is
    // decoded here; what the epilogue and the exchange need is decoded again AFTER the K loop (nothing of it stays live across the loop)
    int gmap0 = 0;
    if (SPLIT && (CDX2_DW(vd, CDX2_W2_XG) & CDX2_XG_GOP)) gmap0 = CDX2_DW(vd, CDX2_W2_GMAP);
    const Geom g{CDX2_DW(vd, CDX2_W2_LCOLS), CDX2_DW(vd, CDX2_W2_CSTRIDE), CDX2_DW(vd, CDX2_W2_OSTRIDE), sstride, L.stage_off,
                 gmap0 & 255, gmap0 >> 8};
    const bool epi_wave = NWV == 4 || SPLIT_T || wave < 4;
    const bool halo_wave = NWV == 4 || SPLIT_T || wave >= 4;

    // epilogue geometry + per-channel parameters: issued now, consumed after the barrier (latency hides behind the K loop)
    const int etid = tid & 255;
    const int grp = etid >> 5, li = etid & 31;
    const int c = mul24i(grp, coutp >> 3) + 4 * (li & ((1 << shift) - 1));
    const int pos0 = li >> shift, pstep = 32 >> shift;
    const int nv = (coutp >> 5) * l_out;
    // this op's epilogue parameters: fetched during the PREVIOUS op (PIPE), or here (consumed after the barrier either way)
    EpiParams P = PIPE ? F.P : load_params<COND, SPLIT_T, CDX2_PARAMS_ALL != 0, MLP, MEMBER>(L, vd, emb_row, emb_tstride, tid, wave, epi_wave);

    // K loop -> staged partial tiles
    const int n_items = CDX2_DW(vd, CDX2_W2_NITEMS);
    // (grouped ops take their own instantiations -- column map, steady-state loop for the 16x16 streams --; the ordinary ops of a
    //  split / grouped kernel run the very K loops of the plain kernels: r4f op profile, +150-200 cycles of item set-up per op otherwise)
    if (SPLIT && gmap0 != 0) {
        if (CDX2_DW(vd, CDX2_W2_MODE) == CDX_MODE_4X4) {
            if (CDX2_DW(vd, CDX2_W2_NT) == 1) conv_kloop<M4, 1, T, NWV, PROF, SPLIT>(g, L.wblob, vd, ops, it, n_items, lds, tf, lane, wave, ring, prof, L.tune);
            else conv_kloop<M4, 2, T, NWV, PROF, SPLIT>(g, L.wblob, vd, ops, it, n_items, lds, tf, lane, wave, ring, prof, L.tune);
        } else {
            conv_kloop<M16, 1, T, NWV, PROF, SPLIT>(g, L.wblob, vd, ops, it, n_items, lds, tf, lane, wave, ring, prof, L.tune);
        }
    } else if (CDX2_DW(vd, CDX2_W2_MODE) == CDX_MODE_4X4) {
        if (CDX2_DW(vd, CDX2_W2_NT) == 1) conv_kloop<M4, 1, T, NWV, PROF>(g, L.wblob, vd, ops, it, n_items, lds, tf, lane, wave, ring, prof, L.tune);
        else conv_kloop<M4, 2, T, NWV, PROF>(g, L.wblob, vd, ops, it, n_items, lds, tf, lane, wave, ring, prof, L.tune);
    } else {
        conv_kloop<M16, 1, T, NWV, PROF>(g, L.wblob, vd, ops, it, n_items, lds, tf, lane, wave, ring, prof, L.tune);
    }
    if (PROF) stamp(prof ? prof + 7 : nullptr, tid);
    // head of the next op's weight stream: flies through the barrier and the epilogue
    // (issued AFTER the partial tiles are staged: sending the eight loads first, while the MFMAs drain, blocks the wave on the
    //  memory pipe for ~350 cycles before it can write its tile -- measured 9 % slower)
    fetch_next(!PARAMS_AFTER_BARRIER);
    if (PROF) stamp(prof ? prof + 1 : nullptr, tid);
    __syncthreads();
    if (PROF) stamp(prof ? prof + 2 : nullptr, tid);
    const EpiParams Pnext = PARAMS_AFTER_BARRIER ? load_params<COND, SPLIT_T, false, MLP, MEMBER>(L, vdn, emb_next, emb_next_tstride, tid, wave, epi_wave)
                                                 : (PIPE ? F.P : P);

    const EpiDesc e = decode_epi<BWD>(vd);
    int xgw = 0, gmap = 0;
    if (SPLIT) {
        int vde = vd;
        asm volatile("" : "+v"(vde));                   // (decode from scratch: see above)
        xgw = CDX2_DW(vde, CDX2_W2_XG);
        // (W2_XG / W2_GMAP alias W2_DST2 / W2_SAVE: an op with backward extras -- the classifier's part of a grouped guided program --
        //  is never cut or exchanged, and those words are what their names say)
        if (BWD && (e.flags & (CDX2_F2_SAVE | CDX2_F2_GNBWD | CDX2_F2_DUAL))) xgw = 0;
        if (xgw & (CDX2_XG_GOP | CDX2_XG_TRAJ)) gmap = CDX2_DW(vde, CDX2_W2_GMAP);
    }
    if (SPLIT && (xgw & CDX2_XG_GOP)) {
        // ---- grouped op (T = 1): half-wave `grp` = trajectory grp / gpm of the group, lane group gx_lo + grp % gpm (gpm = 8 / k is 2 or 4,
        // so the two half-waves of a wave belong to the same trajectory: g_t is wave-uniform).  This half-wave's trajectory: its sub-slot
        // of the destination / residual slots, its columns of the staged tiles (which hold only the member's channels: relative to the
        // first one), K slices `k x l_out` positions apart.  Every half-wave works.
        const int gx_lo = xgw & 255, gpm = ((xgw >> 8) & 255) - gx_lo, grows = gmap >> 8;
        // fast exchange: the epilogue threads publish, the other waves collect meanwhile (collect_fast); X->seq moves on below
        const bool xfast = CDX2_XCHG_FAST && (xgw & CDX2_XG_XCHG) != 0;
        const unsigned xseq = X->seq + 1;
        float* xtile = xfast ? exchange_tile(*X, xseq, __float_as_int(lds[T * tf])) : nullptr;
        if (epi_wave) {
            const int g_t = __builtin_amdgcn_readfirstlane(gpm == 4 ? grp >> 2 : grp >> 1);
            const int cgo = mul24i(gx_lo + (grp & (gpm - 1)), coutp >> 3) + 4 * (li & ((1 << shift) - 1));
            EpiDesc eg = e;
            eg.dst += g_t * grows * e.dstride;
            eg.res += g_t * grows * e.rstride;
            eg.l_out = l_out * X->k;
            const int stage_g = g.stage + g_t * l_out * sstride - gx_lo * (coutp >> 3);
            bool withhold = false;                          // test hook (cdx_unet2_launch.fault): a lost granule on purpose
            if (xfast) {
                const KArg* S0 = kernarg();
                asm volatile("" : "+s"(S0));
                withhold = S0->fault != 0 && X->m == S0->fault - 1;
            }
            const Pub pub{xtile, __uint_as_float(xseq), g_t * l_out, e.coutp, xfast && !withhold};
            if (e.nk == 1) epilogue<1, BWD, COND, MLP, true>(lds, P, eg, stage_g, cgo, pos0, pstep, li, nv, lane, grp, nullptr, &pub);
            else epilogue<2, BWD, COND, MLP, true>(lds, P, eg, stage_g, cgo, pos0, pstep, li, nv, lane, grp, nullptr, &pub);
        }
    } else {
    const int t_lo = SPLIT_T ? (wave >> 2) : 0, t_step = SPLIT_T ? 2 : 1;
#pragma unroll 1
    for (int t = t_lo; t < T; t += t_step) {
        float* tl = lds + t * tf;
        // (a trajectory past the end of the range -- odd count, last workgroup -- computes on its zeroed region; its saved tensors
        //  go to the spare block [batch] of the workspace, never into a real trajectory's block)
        float* ws = BWD ? L.ws + (size_t)(b0 + t < L.traj_first + L.traj_count ? b0 + t : L.batch) * L.ws_floats : nullptr;
        if (COND && emb_tstride != 0 && t != t_lo && epi_wave && (e.flags & CDX2_F2_EMB)) {
            const float* __restrict__ pe = emb_row + t * emb_tstride + CDX2_DW(vd, CDX2_W2_EMB) + c;
            if (e.flags & CDX2_F2_FILM) {
                P.sc = *reinterpret_cast<const f32x4*>(pe);
                P.em = *reinterpret_cast<const f32x4*>(pe + e.coutp);
            } else P.em = *reinterpret_cast<const f32x4*>(pe);
        }
        // (split programs: the half-waves of lane groups that belong to other members sit this op's epilogue out)
        const int xg = SPLIT ? xgw : 0;
        if (epi_wave && (!SPLIT || (xg & 0xffff) == 0 || (grp >= (xg & 255) && grp < ((xg >> 8) & 255)))) {
            if (BWD && (e.flags & CDX2_F2_GNBWD)) {
                if (e.nk == 1) epilogue_bwd<1>(tl, P, e, g.stage, c, pos0, pstep, li, nv, lane, grp, ws);
                else if (e.nk == 2) epilogue_bwd<2>(tl, P, e, g.stage, c, pos0, pstep, li, nv, lane, grp, ws);
                else epilogue_bwd<CDX2_MAX_NK2>(tl, P, e, g.stage, c, pos0, pstep, li, nv, lane, grp, ws);
            } else if (e.nk == 1) epilogue<1, BWD, COND, MLP>(tl, P, e, g.stage, c, pos0, pstep, li, nv, lane, grp, ws);
            else if (e.nk == 2) epilogue<2, BWD, COND, MLP>(tl, P, e, g.stage, c, pos0, pstep, li, nv, lane, grp, ws);
            else epilogue<CDX2_MAX_NK2, BWD, COND, MLP>(tl, P, e, g.stage, c, pos0, pstep, li, nv, lane, grp, ws);
        }
        if (halo_wave && !(SPLIT && (xgw & CDX2_XG_TRAJ))) {
            // wave w (mod 4) rewrites halo row w of the destination (the arena hands this LDS to slots of other shapes in between)
            const int hw = wave & 3;
            const int hrow = hw < CDX2_HALO2 ? hw : e.l_out + hw;
            for (int j = lane * 4; j < e.dstride; j += 256)
                *reinterpret_cast<f32x4*>(tl + e.dst + hrow * e.dstride + j) = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (BWD && (e.flags & CDX2_F2_DUAL))
                for (int j = lane * 4; j < e.d2stride; j += 256)
                    *reinterpret_cast<f32x4*>(tl + e.dst2 + hrow * e.d2stride + j) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    }
    if (SPLIT && halo_wave && (xgw & (CDX2_XG_GOP | CDX2_XG_TRAJ))) {
        // group slots: halo row w of EVERY trajectory's sub-slot (XG_TRAJ: the exchange brings the other trajectories' data rows only;
        // the descriptor's destination is this member's sub-slot)
        const int hw = wave & 3;
        const int hrow = hw < CDX2_HALO2 ? hw : l_out + hw;
        const int hrows = gmap >> 8;
        const int dbase = e.dst - ((xgw & CDX2_XG_TRAJ) ? X->m * hrows * e.dstride : 0);
        for (int tt = 0; tt < X->k; ++tt)
            for (int j = lane * 4; j < e.dstride; j += 256)
                *reinterpret_cast<f32x4*>(lds + dbase + (tt * hrows + hrow) * e.dstride + j) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (PIPE) F.P = Pnext;
    if (SPLIT && CDX2_XCHG_FAST && (xgw & CDX2_XG_GOP) && (xgw & CDX2_XG_XCHG)) {
        // grouped op, fast exchange: the halo waves collect while the epilogue waves still compute and publish
        const unsigned xseq = X->seq + 1;
        if (halo_wave && !CDX2_XCHG_NOWAIT) {
            const int gi = __float_as_int(lds[T * tf]);
            collect_fast(*X, gi, xseq, exchange_tile(*X, xseq, gi), xgw, gmap, lds, e.dst, e.dstride, l_out, e.coutp, tid & 255);
        }
        X->seq = xseq;
    } else
    if (SPLIT && (xgw & CDX2_XG_XCHG)) {
        if (PROF && CDX2_PROF_FETCH == 2) stamp(prof ? prof + 4 : nullptr, tid);
        __syncthreads();                                             // the epilogue's stores to the destination slot are in LDS
        split_exchange<WG<NWV>::THREADS>(*X, __float_as_int(lds[T * tf]), xgw, gmap, lds, e.dst, e.dstride, l_out, e.c_out, e.coutp, tid,
                                         (PROF && CDX2_PROF_FETCH == 2) ? prof : nullptr);
        if (PROF && CDX2_PROF_FETCH == 2) stamp(prof ? prof + 6 : nullptr, tid);
    }
    __syncthreads();
    if (PROF) stamp(prof ? prof + 3 : nullptr, tid);
}

// T = 1: two workgroups per CU must be able to co-reside (that is what hides this latency-bound kernel's stalls from B = 512
// on), i.e. at most 256 VGPR + AGPR per lane -- the second launch-bound argument is waves per SIMD.
// PROF: the s_memtime stamps of tools/op_profile2.py exist only in the instantiations a launch with `prof != NULL` selects -- even
// untaken, their scalar branches and the values they keep alive cost 1-3 % (A/B on MI355X).
// COND: the instantiations that understand conditional requests (per-trajectory FiLM rows, the classifier-free-guidance pair, EDM /
// consistency step kinds).  A separate template parameter so that the unconditional kernels -- the headline path, scalar-register
// bound -- compile to exactly the code they had before.
template <int T, int NWV, bool BWD, bool PROF, bool COND = false, bool MLP = false, bool SPLIT = false>
__global__ __launch_bounds__(NWV * 64, (NWV == 8 || T == 1) ? 2 : 1) void cdx_unet2_kernel(const cdx_unet2_launch L) {
    static_assert(!(COND && BWD), "conditional requests have no backward-op variant");
    static_assert(!MLP || (COND && T == 1 && NWV == 8), "batch-tiled MLP programs: the conditional one-trajectory 8-wave shape");
    // (SPLIT && BWD: the GROUPED GUIDED program -- the denoiser's stream-bound layers grouped, the classifier's forward / backward ops on
    //  the member's own trajectory; round 6)
    static_assert(!SPLIT || (T == 1 && NWV == 8 && !COND), "split / grouped programs: unconditional one-trajectory 8-wave shape");
    constexpr int THREADS = WG<NWV>::THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = L.horizon, D = L.dim, HD = H * D, tf = L.traj_floats;
    // split programs: 8 k consecutive workgroups hold 8 trajectories x k members; the members of a trajectory are 8 workgroups apart,
    // i.e. on the same XCD (workgroup i runs on XCD i % 8)
    if (!SPLIT) {
        // REPAIR launch (cdx_unet2_launch.run_if): enqueued behind a split / grouped launch with the same tensors; it recomputes the
        // request on this ordinary program only if that launch reported a lost granule -- otherwise every workgroup leaves here
        const KArg* S0 = kernarg();
        asm volatile("" : "+s"(S0));
        const int* gate = S0->run_if;
        if (gate != nullptr && __builtin_nontemporal_load(gate) == 0) return;
    }
    XState X{0, 1, 0u, false};
    int grp_idx = 0;
    bool grouped = false;        // grouped program: the k members of a group own k trajectories (one each) instead of one together
    if (SPLIT) {
        // group formation (see XState): a ticket from this XCD's counter -> (group, member).  256 workgroups, 32 per XCD.
        const KArg* S0 = kernarg();
        asm volatile("" : "+s"(S0));
        X.k = S0->split_k;
        grouped = S0->split_group != 0;
        X.seq = S0->xseq0;
        const int per_xcd = (int)gridDim.x >> 3;                                  // workgroups per XCD: 32
        const unsigned xcc = __builtin_amdgcn_s_getreg(CDX2_GETREG_XCC_ID) & 15u;
        unsigned* xtick = reinterpret_cast<unsigned*>(S0->xbuf + (size_t)(gridDim.x / X.k) * 4 * S0->xchg_floats);   // behind the tiles
        if (tid == 0) {
            // (one 128-byte line per counter: only workgroups of XCD `xcc` ever touch line `xcc`)
            const unsigned t = __hip_atomic_fetch_add(xtick + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - S0->xtick0;
            const bool fits = xcc < 8u && t < (unsigned)per_xcd;
            const int g = fits ? (int)xcc * (per_xcd / X.k) + (int)t / X.k : 0;
            lds[T * tf] = __int_as_float(g);
            lds[T * tf + 1] = __int_as_float(fits ? (int)t % X.k : -1);
            // who ended up where (host-side diagnostics; read after the launch): [group][member] = launch tag | XCC id
            if (fits) xtick[16 * 32 + g * X.k + (int)t % X.k] = ((S0->xseq0 + 1u) << 4) | xcc;
        }
        __syncthreads();
        grp_idx = __builtin_amdgcn_readfirstlane(__float_as_int(lds[T * tf]));
        X.m = __builtin_amdgcn_readfirstlane(__float_as_int(lds[T * tf + 1]));
        if (X.m < 0) {                     // more than 32 workgroups on this XCD, or an XCC id past 7: no group for this workgroup
            if (tid == 0) xchg_report(2, X, (int)S0->xseq0, 0, (int)xcc);
            X.m = 0;
            X.dead = true;
        }
    }
    const int moff = SPLIT ? X.m * L.n_ops : 0;           // member m's op i is descriptor m * n_ops + i
    const int b0 = L.traj_first + (SPLIT ? (grouped ? grp_idx * X.k + X.m : grp_idx) : (int)blockIdx.x * T);
    const int b_end = L.traj_first + L.traj_count;      // this launch covers trajectories [traj_first, traj_first + traj_count) of the batch
    unsigned long long* lprof = reinterpret_cast<unsigned long long*>(lds + T * tf + (SPLIT ? 4 : 0));     // (split kernels: the path word first)
    const bool profiling = PROF && L.prof != nullptr && blockIdx.x == 0;
    if (profiling) stamp(lprof + (size_t)L.n_ops * 8, tid);

    // descriptor + first item + weight stream of op 0 first: they fly while the state is set up
    const cint* ops = as_const(L.ops);
    int vd = load_desc<NWV>(L.ops, moff, lane, wave);
    Item it = inline_item(vd);
    Ring<WG<NWV>::PF> ring;
    if (wave < CDX2_DW(vd, CDX2_W2_NITEMS)) prefetch_ring(it, L.wblob, lane, ring);

    // ---- clear the workgroup's LDS once (halo rows and pad channels of the state slots), then load x_T ----
    for (int i = tid * 4; i < T * tf; i += THREADS * 4)
        *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    {
    const KArg* S = kernarg();
    asm volatile("" : "+s"(S));
    // EDM / consistency plans (kinds 5-7; reference newedm.py:130-148): the network sees c_in * x, the authoritative state lives in
    // x_out (as for compact programs) and the LDS slot holds the scaled copy
    const bool edm0 = COND && S->n_steps > 0 && S->edm_plan;
    const float c_in0 = edm0 ? S->steps[0].alpha : 1.0f;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (b0 + t >= b_end) break;
        const size_t xbase = (size_t)(b0 + t) * HD;
        for (int e = tid; e < HD; e += THREADS) {
            const int n = e / D, c = e - n * D;
            float v = S->x_in[xbase + e];
            if (S->init_blend) {
                // x_T = z * temperature, then the fix-mask blend with the prior (reference diffusionsde.py:509-510): the same
                // four roundings as the ATen ops it replaces -- no fma contraction
                v = __fmul_rn(v, S->x_scale);
                if (S->fix_mask) {
                    const float m = S->fix_mask[e];
                    v = __fadd_rn(__fmul_rn(v, __fsub_rn(1.0f, m)), __fmul_rn(S->prior[xbase + e], m));
                }
            }
            lds[t * tf + S->x_off + n * S->x_stride + c] = (COND && edm0) ? c_in0 * v : v;
            if (S->compact || (COND && edm0)) S->x_out[xbase + e] = v;
        }
    }
    }
    __syncthreads();
    // log_p pass only (programs with a classifier head, n_steps == 0 and logp_out given): no forward of the whole op list, just the
    // block after the step loop -- one trajectory per workgroup, FiLM row b of the table for trajectory b (per-sample timesteps)
    // (the flag is re-derived from the kernarg segment where it is needed: one more value live across the op loop is a scalar spill)
    auto logp_only_f = [&]() -> bool {
        if (!BWD) return false;
        const KArg* S0 = kernarg();
        asm volatile("" : "+s"(S0));
        return S0->n_steps == 0 && S0->logp_out != nullptr;
    };
    const int n_iter = logp_only_f() ? 0 : (L.n_steps > 0 ? L.n_steps : 1);
    const int HDp = (HD + 3) & ~3;                     // ws block of a trajectory: [multistep memory / EDM slope | x_old | p_cond]
    constexpr bool PIPE = pipe_params<T, BWD>();        // (see OpFetch)
    constexpr bool SPLIT_T0 = NWV == 8 && T >= 2;
    int n_pass_all = 1;
    if (COND) {
        const KArg* S0 = kernarg();
        asm volatile("" : "+s"(S0));
        n_pass_all = S0->n_pass == 2 ? 2 : 1;
    }
    // FiLM rows of forward (step, pass): one row per step, or one per (step, trajectory) for conditional nets; the second pass of a
    // classifier-free-guidance pair takes the zero-condition table
    auto emb_of = [&](int step, int pass, int& tstride) -> const float* {
        const float* row = L.emb + (size_t)step * L.emb_ld;
        tstride = 0;
        if (COND) {
            const KArg* S0 = kernarg();
            asm volatile("" : "+s"(S0));
            if (pass == 1) row = S0->emb_u + (size_t)step * L.emb_ld;
            else if (S0->emb_per_traj) {      // (the table ends with two spare rows: a half-empty last workgroup reads past its batch)
                row = L.emb + ((size_t)step * L.batch + b0) * L.emb_ld;
                tstride = L.emb_ld;
            }
        }
        return row;
    };
    OpFetch F;
    int vdn_keep = 0;
    if (PIPE) {
        int ts0;
        const float* row0 = emb_of(0, 0, ts0);
        F.P = load_params<COND, SPLIT_T0, false, MLP>(L, vd, row0, ts0, tid, wave, NWV == 4 || SPLIT_T0 || wave < 4);
        vdn_keep = load_desc<NWV>(L.ops, moff + (L.n_ops > 1 ? 1 : 0), lane, wave);
    }
    for (int step = 0; step < n_iter; ++step) {
      const int n_pass = n_pass_all;
#pragma unroll 1
      for (int pass = 0; pass < n_pass; ++pass) {
        int emb_tstride, emb_next_tstride;
        const float* __restrict__ emb_row = emb_of(step, pass, emb_tstride);
        // the forward after this one (its first op's parameters are fetched during this forward's last op)
        const bool last_fwd = pass + 1 == n_pass && step + 1 >= n_iter;
        const float* __restrict__ emb_fwd_next = last_fwd ? emb_of(step, pass, emb_next_tstride)
                                                          : emb_of(pass + 1 < n_pass ? step : step + 1, pass + 1 < n_pass ? pass + 1 : 0, emb_next_tstride);
        const int ts_fwd_next = emb_next_tstride;
        for (int oi = 0; oi < L.n_ops; ++oi) {
            // next op's descriptor (the last op fetches op 0 of the next step): one coalesced load, needed after the K loop
            // (PIPE: it was fetched during the previous op; this op fetches the one after it)
            const int vdn = PIPE ? vdn_keep : load_desc<NWV>(L.ops, moff + (oi + 1 < L.n_ops ? oi + 1 : 0), lane, wave);
            // (profile the SECOND forward when there is one: instruction / scalar caches warm, like every later step)
            unsigned long long* pslot = (profiling && step == (L.n_steps > 1 ? 1 : 0)) ? lprof + (size_t)oi * 8 : nullptr;
            if (PROF) stamp(pslot, tid);
            const bool wrap = oi + 1 >= L.n_ops;
            int on2 = oi + 2;
            if (on2 >= L.n_ops) on2 -= L.n_ops;
            if (on2 >= L.n_ops) on2 = 0;
            // (MLP programs: `pass` tells the context-slot op what to load -- 0 the condition, 1 zeros (unconditional forward of a
            //  pair), 2 nothing: one forward per step and the slot was filled by step 0)
            // (grouped GUIDED programs: running the classifier's ops -- ordinary ops in every member's view -- on the instantiation without
            //  member / exchange code changes nothing: 9.60-9.68 ms either way, profiles/r06_guided_group_ab.txt)
            run_op<T, NWV, BWD, PROF, COND, MLP, SPLIT, SPLIT>(L, ops, vd, vdn, it, emb_row, emb_tstride, lds, tid, ring, pslot, b0, F,
                                                               wrap ? emb_fwd_next : emb_row, wrap ? ts_fwd_next : emb_tstride, on2,
                                                               (MLP && n_pass == 1 && step > 0) ? 2 : pass, &X);
            vd = vdn;
            if (PIPE) vdn_keep = F.vdn2;
        }
        if (COND && n_pass == 2 && pass == 0) {
            // conditional prediction -> the trajectory's ws block; a compact program's state slot was arena memory during the
            // forward: rebuild it from x_out for the second forward
            const KArg* S = kernarg();
            asm volatile("" : "+s"(S));
#pragma unroll 1
            for (int t = 0; t < T; ++t) {
                if (b0 + t >= b_end) break;
                const size_t xbase = (size_t)(b0 + t) * HD;
                float* tl = lds + t * tf;
                float* wsb = S->ws + (size_t)(b0 + t) * S->ws_floats;
                const float c_in = S->edm_plan ? S->steps[step].alpha : 1.0f;
                for (int e = tid; e < HD; e += THREADS) {
                    const int n = e / D, c = e - n * D;
                    wsb[2 * HDp + e] = tl[S->pred_off + n * S->pred_stride + c];
                    if (S->compact) tl[S->x_off + n * S->x_stride + c] = c_in * S->x_out[xbase + e];
                }
                if (S->compact) {
                    const int xs = S->x_stride, xb = S->x_off - CDX2_HALO2 * xs;
                    for (int i = tid; i < (H + 2 * CDX2_HALO2) * xs; i += THREADS) {
                        const int r = i / xs, c = i - r * xs;
                        if (r < CDX2_HALO2 || r >= H + CDX2_HALO2 || c >= D) tl[xb + i] = 0.f;
                    }
                }
            }
            __syncthreads();
        }
      }
        if (L.n_steps == 0) break;
        // The solver step's pointers and offsets are read from the kernarg segment HERE, through a pointer the optimiser cannot
        // see through across iterations: as by-value kernel arguments they would stay live in ~35 SGPRs over the whole op loop,
        // where the descriptor fields already fill the scalar register file (each spill / reload is a VALU v_writelane /
        // v_readlane on the critical path of every op).
        const KArg* S = kernarg();
        asm volatile("" : "+s"(S));
        // ---- clip, eps/x0 conversion, solver update, fix-mask blend on the LDS-resident state (kinds 0-4) ----
        const cdx_step st = S->steps[step];
        const float al = st.alpha, sg = st.sigma;
        const float k0 = st.k[0], k1 = st.k[1], k2 = st.k[2], k3 = st.k[3], k4 = st.k[4];
        const bool edm = COND && st.kind >= 5;
        const bool xglob = S->compact || edm;                  // the authoritative state lives in x_out, memory in the ws block
        const float c_next = (edm && step + 1 < S->n_steps) ? S->steps[step + 1].alpha : 1.0f;   // the NEXT forward sees c_in * x
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if (b0 + t >= b_end) break;
            const int b = b0 + t;
            const size_t xbase = (size_t)b * HD;
            float* tl = lds + t * tf;
            float* wsb = (COND && (xglob || n_pass == 2)) ? S->ws + (size_t)b * S->ws_floats : nullptr;
            for (int e = tid; e < HD; e += THREADS) {
                const int n = e / D, c = e - n * D;
                const int xo = S->x_off + n * S->x_stride + c;
                // compact programs: the authoritative state lives in x_out (global), the LDS slot only feeds op 0 of the next forward
                const float x = xglob ? S->x_out[xbase + e] : tl[xo];
                float p = tl[S->pred_off + n * S->pred_stride + c];
                if (COND && n_pass == 2) p = S->cfg_w * wsb[2 * HDp + e] + (1.0f - S->cfg_w) * p;   // w * cond + (1 - w) * uncond
                if (edm) {
                    // EDM (reference newedm.py:387-401, legacy edm.py:118-160): D = clip(c_skip x + c_out F), slope = (x - D) / sigma;
                    // kind 7 = consistency model (consistency_model.py:412-427): x <- f(x) [mask], then re-noise for the next level
                    float dn = k0 * x + k1 * p, xe;
                    if (S->x_min) dn = fmaxf(dn, S->x_min[e]);
                    if (S->x_max) dn = fminf(dn, S->x_max[e]);
                    if (st.kind == 7) {
                        xe = dn;
                        if (S->fix_mask) {
                            const float m = S->fix_mask[e];
                            xe = xe * (1.0f - m) + S->prior[xbase + e] * m;
                        }
                        if (st.noise_idx >= 0) xe += k3 * S->noise[((size_t)st.noise_idx * S->batch + b) * HD + e];
                    } else {
                        const float sl = (x - dn) / k2;
                        if (st.kind == 5) {
                            xe = x - sl * k3;
                            if (st.push) { wsb[e] = sl; wsb[HDp + e] = x; }
                        } else {
                            xe = wsb[HDp + e] - (wsb[e] + sl) / 2.0f * k3;
                        }
                        if (S->fix_mask) {
                            const float m = S->fix_mask[e];
                            xe = xe * (1.0f - m) + S->prior[xbase + e] * m;
                        }
                    }
                    tl[xo] = c_next * xe;
                    S->x_out[xbase + e] = xe;
                    continue;
                }
                // classifier guidance (reference diffusionsde.py:153-173): the prediction is shifted along d log p / d x_t BEFORE
                // it is clipped; cg_scale[step] = -w sigma (noise prediction) or w sigma^2 / alpha (x0 prediction), frozen by the host
                if (BWD && S->cg_scale) p += S->cg_scale[step] * tl[S->grad_off + n * S->grad_stride + c];
                if (S->predict_noise) {
                    if (S->x_max) p = fmaxf(p, (x - al * S->x_max[e]) / sg);
                    if (S->x_min) p = fminf(p, (x - al * S->x_min[e]) / sg);
                } else {
                    if (S->x_min) p = fmaxf(p, S->x_min[e]);
                    if (S->x_max) p = fminf(p, S->x_max[e]);
                }
                float eps, xth, xn;
                if (S->predict_noise) {
                    eps = p; xth = (x - sg * p) / al;
                } else {
                    xth = p; eps = (x - al * p) / sg;
                }
                if (st.kind >= 3) {
                    // legacy DDPM class (reference diffusion/ddpm.py:153-164, 230-241): fix-mask on the prediction
                    const float m = S->fix_mask ? S->fix_mask[e] : 0.f;
                    if (st.kind == 3) {
                        p = p * (1.0f - m);
                        xn = k0 * (x - k1 * p);
                    } else {
                        p = p * (1.0f - m) + x * m;
                        xn = k0 * (k1 * x + k2 * p);
                    }
                    if (st.noise_idx >= 0) xn += k3 * S->noise[((size_t)st.noise_idx * S->batch + b) * HD + e];
                } else if (st.kind == 0) {
                    xn = k0 * (x - k1 * eps) + k2 * eps;
                    if (st.noise_idx >= 0) xn += k3 * S->noise[((size_t)st.noise_idx * S->batch + b) * HD + e];
                } else if (st.kind == 1) {
                    xn = k0 * ((x - k1 * eps) / k2) + k3 * eps;
                } else {
                    if (st.flags & CDX_STEP_MASK_PRED) {     // legacy DPMSolver (dpmsolver.py:257-264)
                        const float m = S->fix_mask ? S->fix_mask[e] : 0.f;
                        eps = eps * (1.0f - m);
                        xth = xth * (1.0f - m) + x * m;
                    }
                    float v = (st.vsel & 1) ? xth : eps;
                    if (st.vsel == 2) v = k3 * xth - k4 * (S->compact ? S->ws[(size_t)b * S->ws_floats + e] : tl[S->prev_off + e]);
                    if (st.vsel == 3) v = k3 * eps - k4 * (S->compact ? S->ws[(size_t)b * S->ws_floats + e] : tl[S->prev_off + e]);
                    xn = k0 * x - k1 * v;
                    if (st.noise_idx >= 0) xn += k2 * S->noise[((size_t)st.noise_idx * S->batch + b) * HD + e];
                }
                if (S->fix_mask) {
                    const float m = S->fix_mask[e];
                    xn = xn * (1.0f - m) + S->prior[xbase + e] * m;
                }
                if (st.push) {
                    if (S->compact) S->ws[(size_t)b * S->ws_floats + e] = st.push == 2 ? eps : xth;
                    else tl[S->prev_off + e] = st.push == 2 ? eps : xth;
                }
                tl[xo] = xn;
                if (S->compact) S->x_out[xbase + e] = xn;
            }
            if (S->compact) {
                // the LDS state slot of a compact program is an arena slot: other tensors lived there during the forward, so its halo
                // rows and pad channels are rewritten with the zeros every conv source needs
                const int xs = S->x_stride, xb = S->x_off - CDX2_HALO2 * xs;
                for (int i = tid; i < (H + 2 * CDX2_HALO2) * xs; i += THREADS) {
                    const int r = i / xs, c = i - r * xs;
                    if (r < CDX2_HALO2 || r >= H + CDX2_HALO2 || c >= D) tl[xb + i] = 0.f;
                }
            }
        }
        __syncthreads();
    }
    const bool logp_only = logp_only_f();
    if (BWD && (L.n_steps > 0 || logp_only)) {
        // final log_p (reference diffusionsde.py:597-601): the classifier's forward ops once more, on the final state, timestep 0
        // (log_p pass only: on the state just loaded, the trajectory's own row of the table)
        const KArg* S = kernarg();
        asm volatile("" : "+s"(S));
        if (S->logp_out != nullptr) {
            const int first = S->logp_first_op, head = S->logp_head_op;
            const float* __restrict__ emb_row = L.emb + (size_t)(logp_only ? b0 : L.n_steps) * L.emb_ld;
            // (grouped guided programs: the member's own descriptors -- the classifier's ops are ordinary ops in every member's view)
            vd = load_desc<NWV>(L.ops, moff + first, lane, wave);
            it = inline_item(vd);
            if (wave < CDX2_DW(vd, CDX2_W2_NITEMS)) prefetch_ring(it, L.wblob, lane, ring);
            for (int oi = first; oi < head; ++oi) {
                const int vdn = load_desc<NWV>(L.ops, moff + oi + 1, lane, wave);
                run_op<T, NWV, BWD, PROF, COND>(L, ops, vd, vdn, it, emb_row, 0, lds, tid, ring, nullptr, b0, F, emb_row, 0, 0);
                vd = vdn;
            }
#pragma unroll 1
            for (int t = 0; t < T; ++t)
                run_head_fwd<THREADS>(L, vd, emb_row, lds + t * tf, tid, b0 + t < b_end ? S->logp_out + (b0 + t) : nullptr);
        }
    }
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (b0 + t >= b_end) break;
        const size_t xbase = (size_t)(b0 + t) * HD;
        // one forward (n_steps == 0): the network output -- of a program with backward ops, the gradient slot
        const KArg* S = kernarg();
        asm volatile("" : "+s"(S));
        const bool want_grad = BWD && S->n_steps == 0 && S->grad_off >= 0 && S->with_backward;
        const int off = S->n_steps == 0 ? (want_grad ? S->grad_off : S->pred_off) : S->x_off;
        const int str = S->n_steps == 0 ? (want_grad ? S->grad_stride : S->pred_stride) : S->x_stride;
        float* __restrict__ xo = S->x_out;
        if ((S->compact || (COND && S->edm_plan)) && S->n_steps > 0) break;            // the state is already there
        // (split programs: a thread that gave up on a granule makes its workgroup store NaN -- a failed exchange must never look
        //  like a sample; the error word tells the host)
        const bool poisoned = SPLIT && __syncthreads_or(X.dead ? 1 : 0) != 0;
        if (SPLIT && !grouped && X.m != 0) break;                                      // every member holds the result: member 0 stores it
        if (BWD && logp_only) break;                                                   // nothing but logp_out is produced
        for (int e = tid; e < HD; e += THREADS) {
            const int n = e / D, c = e - n * D;
            xo[xbase + e] = (SPLIT && poisoned) ? __builtin_nanf("") : lds[t * tf + off + n * str + c];
        }
    }
    if (profiling) {
        stamp(lprof + (size_t)L.n_ops * 8 + 1, tid);
        __syncthreads();
        for (int i = tid; i < L.n_ops * 8 + 2; i += THREADS) L.prof[i] = lprof[i];
    }
}

// FiLM table: one workgroup per step record.  Linear -> Mish -> Linear -> Mish -> stacked per-block Linear, fp32 fma chains.
__global__ __launch_bounds__(256) void cdx_unet2_embtab_kernel(const cdx_unet2_embtab_args A) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float* v0 = sh;
    float* h = v0 + A.emb_dim;
    float* m = h + A.hidden;
    float* raw = m + A.md;
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < A.emb_dim; i += 256) v0[i] = A.temb[(size_t)r * A.emb_dim + i];
    __syncthreads();
    for (int o = tid; o < A.hidden; o += 256) {
        float acc = A.wblob[A.b0 + o];
        for (int i = 0; i < A.emb_dim; ++i) acc = fmaf(A.wblob[A.w0 + (size_t)i * A.hidden + o], v0[i], acc);
        h[o] = mish2(acc);
    }
    __syncthreads();
    for (int o = tid; o < A.md; o += 256) {
        float acc = A.wblob[A.b2 + o];
        for (int i = 0; i < A.hidden; ++i) acc = fmaf(A.wblob[A.w2 + (size_t)i * A.md + o], h[i], acc);
        raw[o] = acc;
        m[o] = mish2(acc);
    }
    __syncthreads();
    float* orow = A.out + (size_t)r * A.out_ld;
    for (int o = tid; o < A.n_emb; o += 256) {
        float acc = A.wblob[A.b3 + o];
        for (int i = 0; i < A.md; ++i) acc = fmaf(A.wblob[A.w3 + (size_t)i * A.n_emb + o], m[i], acc);
        orow[A.col0 + o] = acc;
    }
    for (int o = tid; o < A.n_raw; o += 256) {            // rows applied to the RAW embedding (classifier head)
        float acc = A.wblob[A.b4 + o];
        for (int i = 0; i < A.md; ++i) acc = fmaf(A.wblob[A.w4 + (size_t)i * A.n_raw + o], raw[i], acc);
        orow[A.col4 + o] = acc;
    }
}

}  // namespace

extern "C" {

int cdx_unet2_embtab(const cdx_unet2_embtab_args* A, void* hip_stream) {
    cdx_set_err("");
    if (!A || !A->wblob || !A->temb || !A->out) { cdx_set_err("null pointer in embtab args"); return CDX_EINVAL; }
    if (A->n_rows == 0) return CDX_OK;
    if (A->n_rows < 0 || A->emb_dim <= 0 || A->hidden <= 0 || A->md <= 0 || A->n_emb <= 0 || A->n_raw < 0) { cdx_set_err("non-positive size"); return CDX_EINVAL; }
    if (A->out_ld < A->col0 + A->n_emb || (A->n_raw > 0 && A->out_ld < A->col4 + A->n_raw) || A->col0 < 0 || A->col4 < 0) {
        cdx_set_err("embedding table columns do not fit the row stride"); return CDX_EINVAL;
    }
    const size_t sh = (size_t)(A->emb_dim + A->hidden + 2 * A->md) * sizeof(float);
    if (sh > 64u * 1024u) { cdx_set_err("embedding MLP too wide for the table kernel"); return CDX_EINVAL; }
    hipLaunchKernelGGL(cdx_unet2_embtab_kernel, dim3(A->n_rows), dim3(256), sh, reinterpret_cast<hipStream_t>(hip_stream), *A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int cdx_unet2_run(const cdx_unet2_launch* L, void* hip_stream) {
    cdx_set_err("");
    if (!L || !L->ops || !L->wblob || !L->x_in || !L->x_out || !L->emb) { cdx_set_err("null pointer in launch"); return CDX_EINVAL; }
    if (L->batch == 0) return CDX_OK;
    cdx_unet2_launch whole;
    if (L->traj_first == 0 && L->traj_count == 0) {      // range left zero: the whole batch
        whole = *L;
        whole.traj_count = L->batch;
        L = &whole;
    }
    if (L->traj_first < 0 || L->traj_count < 0 || L->traj_first + L->traj_count > L->batch) { cdx_set_err("trajectory range outside the batch"); return CDX_EINVAL; }
    if (L->n_ops <= 0 || L->batch < 0 || L->horizon <= 0 || L->dim <= 0 || L->traj_floats <= 0) { cdx_set_err("non-positive size"); return CDX_EINVAL; }
    if (L->traj_per_wg < 1 || L->traj_per_wg > 3) { cdx_set_err("traj_per_wg must be 1, 2 or 3"); return CDX_EINVAL; }
    if (L->traj_per_wg == 3 && (L->n_waves != 8 || !L->compact)) { cdx_set_err("three trajectories per workgroup: 8-wave compact programs only"); return CDX_EINVAL; }
    if (L->compact && L->x_out == L->x_in) { cdx_set_err("compact program: x_out holds the state during the launch and must not alias x_in"); return CDX_EINVAL; }
    if (L->compact && L->n_steps > 0 && (!L->ws || L->ws_floats < L->horizon * L->dim)) { cdx_set_err("compact program: ws (multistep memory) missing"); return CDX_EINVAL; }
    if (L->n_pass != 0 && L->n_pass != 1 && L->n_pass != 2) { cdx_set_err("n_pass must be 1 or 2"); return CDX_EINVAL; }
    if (L->n_pass == 2 && (!L->emb_u || L->n_steps == 0)) { cdx_set_err("classifier-free-guidance pair: emb_u and a sampling loop required"); return CDX_EINVAL; }
    if ((L->n_pass == 2 || L->edm_plan) && (!L->ws || L->ws_floats < 3 * ((L->horizon * L->dim + 3) & ~3))) {
        cdx_set_err("CFG pair / EDM step kinds: ws with 3 * round4(horizon * dim) floats per trajectory required"); return CDX_EINVAL;
    }
    if (L->edm_plan && L->x_out == L->x_in) { cdx_set_err("EDM step kinds: x_out holds the state during the launch and must not alias x_in"); return CDX_EINVAL; }
    if ((L->n_pass == 2 || L->edm_plan || L->emb_per_traj) && (L->cg_scale != nullptr || L->with_backward != 0)) {
        cdx_set_err("conditional / EDM requests are not available for programs with backward ops"); return CDX_EINVAL;
    }
    if (L->n_waves != 4 && L->n_waves != 8) { cdx_set_err("n_waves must be 4 or 8 (the program is compiled for one of them)"); return CDX_EINVAL; }
    if (L->n_steps > 0 && !L->steps) { cdx_set_err("steps == NULL with n_steps > 0"); return CDX_EINVAL; }
    if (L->n_steps < 0) { cdx_set_err("negative n_steps"); return CDX_EINVAL; }
    if (L->fix_mask && !L->prior) { cdx_set_err("fix_mask given without prior"); return CDX_EINVAL; }
    if ((L->traj_floats | L->x_off | L->x_stride | L->pred_off | L->pred_stride | L->prev_off | L->stage_off | L->emb_ld) & 3) {
        cdx_set_err("LDS offsets/strides and emb_ld must be multiples of 4 floats"); return CDX_EINVAL;
    }
    size_t lds_bytes = (size_t)L->traj_floats * L->traj_per_wg * sizeof(float);
    if (L->split_k != 0) {
        lds_bytes += 16;                            // group index / member index words of split / grouped launches (the kernel's prologue)
        if (lds_bytes < 96u * 1024u) lds_bytes = 96u * 1024u;       // more than half a CU's LDS: ONE workgroup per CU, 32 per XCD
    }
    if (L->prof) lds_bytes += (size_t)(L->n_ops * 8 + 2) * sizeof(unsigned long long);
    if (lds_bytes > 160u * 1024u) { cdx_set_err("program needs more than 160 KiB of LDS"); return CDX_ELDS; }
    const bool guided = L->cg_scale != nullptr || L->with_backward != 0;
    if (guided && L->n_waves != 8) { cdx_set_err("programs with backward ops (classifier guidance) run in the 8-wave shape only"); return CDX_EINVAL; }
    if (guided && L->ws_floats > 0 && !L->ws) { cdx_set_err("program keeps saved tensors in a global workspace: ws == NULL"); return CDX_EINVAL; }
    if (L->ws_floats < 0 || (L->ws_floats & 3)) { cdx_set_err("ws_floats must be a non-negative multiple of 4"); return CDX_EINVAL; }
    if (L->cg_scale && (L->grad_off < 0 || (L->grad_off & 3) || (L->grad_stride & 3))) { cdx_set_err("cg_scale given without a gradient slot"); return CDX_EINVAL; }
    if (L->logp_out && (!guided || L->logp_first_op < 0 || L->logp_head_op <= L->logp_first_op || L->logp_head_op >= L->n_ops)) {
        cdx_set_err("logp_out: programs with a classifier head only, with 0 <= logp_first_op < logp_head_op < n_ops"); return CDX_EINVAL;
    }
    if (L->logp_out && L->n_steps == 0 && (L->traj_per_wg != 1 || L->compact)) {
        cdx_set_err("logp_out without a sampling loop (log_p pass of a batch): one trajectory per workgroup, state in LDS"); return CDX_EINVAL;
    }
    void (*kern)(const cdx_unet2_launch);
    const bool cond = L->n_pass == 2 || L->edm_plan || L->emb_per_traj;
    int split_grid = 0;
    if (L->split_k != 0 && L->run_if) { cdx_set_err("run_if: the repair launch is an ORDINARY launch (split_k == 0)"); return CDX_EINVAL; }
    if (L->fault < 0 || (L->fault != 0 && L->split_k == 0)) { cdx_set_err("fault: test hook of split / grouped launches"); return CDX_EINVAL; }
    if (L->split_k != 0) {
        if ((L->split_k != 2 && L->split_k != 4) || cond || L->mlp || L->traj_per_wg != 1 || L->n_waves != 8 || L->compact ||
            !L->xbuf || !L->xerr || L->xchg_floats <= 0 || (L->xchg_floats & 3)) {
            cdx_set_err("split / grouped program: split_k 2 or 4, one trajectory per workgroup, 8 waves, unconditional, xbuf / xerr given"); return CDX_EINVAL;
        }
        if (guided && L->prof) { cdx_set_err("split / grouped programs with backward ops: no op profiling"); return CDX_EINVAL; }
        // split: split_k workgroups per trajectory; grouped: split_k trajectories per group of split_k workgroups.  ALWAYS one workgroup
        // per CU of the whole chip -- 256, 32 per XCD, all resident: that is what lets the workgroups form their groups from per-XCD
        // tickets (see XState); groups past the batch compute on zeros
        const int groups_needed = L->split_group ? (L->traj_count + L->split_k - 1) / L->split_k : L->traj_count;
        split_grid = CDX2_N_CUS;
        if (groups_needed > split_grid / L->split_k) {
            cdx_set_err("split / grouped program: at most 256 / split_k groups per launch (every workgroup must be resident)"); return CDX_EINVAL;
        }
        kern = guided ? cdx_unet2_kernel<1, 8, true, false, false, false, true>
                      : L->prof ? cdx_unet2_kernel<1, 8, false, true, false, false, true> : cdx_unet2_kernel<1, 8, false, false, false, false, true>;
    } else if (L->split_group) {
        cdx_set_err("split_group without split_k"); return CDX_EINVAL;
    } else if (L->mlp) {
        if (guided || L->traj_per_wg != 1 || L->n_waves != 8 || L->emb_per_traj || L->compact || L->prof) {
            cdx_set_err("batch-tiled MLP program: one tile per workgroup, 8 waves, one table row per step, no backward ops"); return CDX_EINVAL;
        }
        kern = cdx_unet2_kernel<1, 8, false, false, true, true>;
    } else if (cond) {
        if (L->prof) { cdx_set_err("op profiling: unconditional requests only"); return CDX_EINVAL; }
        kern = L->n_waves == 8 ? (L->traj_per_wg == 3 ? cdx_unet2_kernel<3, 8, false, false, true>
                                  : L->traj_per_wg == 2 ? cdx_unet2_kernel<2, 8, false, false, true> : cdx_unet2_kernel<1, 8, false, false, true>)
                               : (L->traj_per_wg == 2 ? cdx_unet2_kernel<2, 4, false, false, true> : cdx_unet2_kernel<1, 4, false, false, true>);
    } else if (L->prof) {                                       // profiling builds of the shapes tools/op_profile2*.py look at
        if (L->n_waves != 8) { cdx_set_err("op profiling: 8-wave shapes only"); return CDX_EINVAL; }
        if (guided && L->traj_per_wg == 3) { cdx_set_err("op profiling of guided programs: one or two trajectories per workgroup"); return CDX_EINVAL; }
        kern = guided ? (L->traj_per_wg == 2 ? cdx_unet2_kernel<2, 8, true, true> : cdx_unet2_kernel<1, 8, true, true>)
                      : (L->traj_per_wg == 3 ? cdx_unet2_kernel<3, 8, false, true>
                         : L->traj_per_wg == 2 ? cdx_unet2_kernel<2, 8, false, true> : cdx_unet2_kernel<1, 8, false, true>);
    } else {
        kern = guided ? (L->traj_per_wg == 3 ? cdx_unet2_kernel<3, 8, true, false>
                         : L->traj_per_wg == 2 ? cdx_unet2_kernel<2, 8, true, false> : cdx_unet2_kernel<1, 8, true, false>)
             : L->n_waves == 8 ? (L->traj_per_wg == 3 ? cdx_unet2_kernel<3, 8, false, false>
                                  : L->traj_per_wg == 2 ? cdx_unet2_kernel<2, 8, false, false> : cdx_unet2_kernel<1, 8, false, false>)
                               : (L->traj_per_wg == 2 ? cdx_unet2_kernel<2, 4, false, false> : cdx_unet2_kernel<1, 4, false, false>);
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    const int grid = split_grid ? split_grid : (L->traj_count + L->traj_per_wg - 1) / L->traj_per_wg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(L->n_waves * 64), lds_bytes, reinterpret_cast<hipStream_t>(hip_stream), *L);
    e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

}  // extern "C"
