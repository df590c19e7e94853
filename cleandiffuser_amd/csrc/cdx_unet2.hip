// cdx_unet2.hip -- second-generation fused "program" kernel for 1-D temporal U-Net denoisers on gfx950 (MI355X / CDNA4).
//
// Replaces, like cdx_unet1d.hip, the ~170 ATen launches per denoiser forward plus the solver arithmetic of the reference loop
// (cleandiffuser/diffusion/diffusionsde.py:526-594 driving cleandiffuser/nn_diffusion/jannerunet.py:154-201) with ONE launch per
// sample() call.  What changed against the first kernel is the per-op fixed cost (round-1 profile: 60 % of the time):
//
//   * 4 wave64 per workgroup, one per SIMD; the output of a conv is cut into (row tile x column group) tiles, one per wave (K is
//     split over spare waves only when a layer has fewer than four tiles); work items are 8-word records read with s_load;
//   * layer descriptors are read with scalar loads straight into SGPRs (no LDS copy, no v_readlane decode);
//   * the ResidualBlock's 1x1 skip conv (jannerunet.py:58, :69) is a plain op whose epilogue adds into the block's output slot;
//   * the per-block FiLM vectors are per-STEP constants: they come from a (steps, n_emb) table (cdx_unet2_embtab_kernel below) as
//     float4 epilogue parameters -- the embedding MLP is gone from the per-forward program;
//   * epilogue: 32 lanes per GroupNorm group, one float4 of consecutive channels per lane, statistics in registers (two-pass),
//     Mish, + FiLM, + residual slot, one float4 LDS store; every per-channel parameter is fetched BEFORE the K loop;
//   * 16 weight records (16 KiB) in flight per wave, and the head of the NEXT op's stream is issued before this op's barrier;
//   * T = 1 or 2 trajectories per workgroup: each streamed weight record feeds T x the MFMAs (B >= 512 halves the L2->CU
//     stream per trajectory, which is what bounds the 1.3-MB layers at L = 4).
//
// Executable specification / CPU twin: oracle/lane_sim2.py.  Program format: engine/program2.py, csrc/cdx_ops2.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/cdx.h"
#include "cdx_ops.h"
#include "cdx_ops2.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Descriptor / item words are read through the constant address space: a wave-uniform address there is an s_load (scalar
// cache, SGPR destination) -- no vector load + v_readfirstlane decode.
typedef const int __attribute__((address_space(4))) cint;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
static __device__ __forceinline__ const cint* as_const(const int* p) { return (const cint*)p; }
#pragma clang diagnostic pop

void cdx_set_err(const char* msg);          // cdx_unet1d.hip

#define NW CDX2_NW2
#define THREADS (NW * 64)
#define PF CDX2_RING2

namespace {

__device__ __forceinline__ float mish2(float x) {
    // x * tanh(softplus(x)), tanh(log(1+e^x)) = n / (n + 2), n = e^x (e^x + 2); softplus threshold 20 as ATen
    const float e = __expf(fminf(x, 20.0f));
    const float n = e * (e + 2.0f);
    return x > 20.0f ? x : x * n * __builtin_amdgcn_rcpf(n + 2.0f);
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

struct Seg { int src, sstr, ccn, taps, pad; };

struct Geom {                    // wave-uniform description of one conv op's K loop
    int l_out, l_in, cstride, transposed, zrow;
    const cint* segs;            // the op's segment table (descriptor words CDX2_W2_SEG0...), re-read at a segment change
    int sstride, stage;
};

struct Cursor { int si, tap, cc; Seg s; };

// Segment `si` of the op (a K segment = one source slot of a channel concat): scalar loads from the descriptor (scalar-cache
// hits; it changes at most once per item).  Kept as loads on purpose: selecting between register-resident structs makes the
// compiler build a scratch array.
__device__ __forceinline__ Seg pick(const Geom& g, int si) {
    const cint* s = g.segs + si * CDX2_SEG2_WORDS;
    return Seg{s[CDX2_S2_SRC], s[CDX2_S2_STRIDE], s[CDX2_S2_CCN], s[CDX2_S2_TAPS], s[CDX2_S2_PAD]};
}

// Input row feeding output position `pos` at tap `tap`, or -1 (outside [0, l_in), odd phase of the stride-2 transposed conv,
// column past l_out): such lanes read the trajectory's all-zero row, so the B fetch needs no predicate.
__device__ __forceinline__ int conv_row(const Geom& g, int pos, int tap, int pad) {
    const int fwd = pos * g.cstride + tap - pad;
    const int num = pos + pad - tap;
    const int bwd = (num & 1) ? -1 : (num >> 1);
    const int q = g.transposed ? bwd : fwd;
    return (pos < g.l_out && q >= 0 && q < g.l_in) ? q : -1;
}

template <class M, int NT>
__device__ __forceinline__ void lane_rows(const Geom& g, const Cursor& c, int col0, int lane, int (&roff)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = conv_row(g, col0 + nt * M::COLS + M::col(lane), c.tap, c.s.pad);
        const int in_slot = __mul24(q, c.s.sstr);                                  // 24-bit multiply: cheap enough to stay branch-free
        roff[nt] = (q >= 0 ? in_slot : g.zrow - c.s.src) + M::koff(lane);          // relative to the segment's source slot
    }
}

template <class M, int NT, int T>
__device__ __forceinline__ void fetch_b(const float* __restrict__ lds, int tf, const Cursor& c, const int (&roff)[NT],
                                        f32x4 (&bv)[T][NT]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            bv[t][nt] = *reinterpret_cast<const f32x4*>(lds + t * tf + c.s.src + roff[nt] + c.cc * M::KSTEP);
}

// ---- weight ring -------------------------------------------------------------------------------------------------------
// PF = 16 1-KiB records in flight per wave, held as RS = 8 slots of SUB = 2 records.  The slot count matters: hipcc keeps
// *counted* `s_waitcnt vmcnt(N)` waits only for register rings unrolled <= 8 deep (measured on this toolchain: 9, 10, 12 and
// 16 slots all degenerate to a full drain per revolution, i.e. one L2 round trip every PF records), while 8 slots x 2 loads
// give the same 16 loads in flight with waits vmcnt(14)/(15).
#define RS 8
#define SUB (PF / RS)
struct Ring { f32x4 rec[RS][SUB]; };  // head of this wave's next weight stream, issued one op ahead

// Profiling stamps go to LDS (a global store would be waited on by the next vmcnt wait and distort the phase being measured).
__device__ __forceinline__ void stamp(unsigned long long* slot, int tid) {
    if (slot && tid == 0) *slot = __builtin_amdgcn_s_memtime();
}

// K loop of one conv op for this wave: items wave, wave + 4, ...; an item = (row tile, column group, K slice).
template <class M, int NT, int T>
__device__ __forceinline__ void conv_kloop(const Geom& g, const float* __restrict__ wblob, const cint* items,
                                           int n_items, float* __restrict__ lds, int tf, int lane, int wave, Ring& ring,
                                           unsigned long long* prof) {
    const int ptid = (wave == 0 && lane == 0) ? 0 : 1;          // stamp() fires for tid == 0 only
    // operand ring depth: 8 when a chunk is only 4 short MFMAs (32 cycles), else 4
    constexpr int BD = (M::KSTEP == 4 && NT * T == 1) ? 8 : 4;
    static_assert(PF % BD == 0, "operand ring must divide the weight ring");
    for (int item = wave; item < n_items; item += NW) {
        const cint* it = items + item * CDX2_ITEM_WORDS;                     // wave-uniform address -> scalar loads
        const int woff = it[CDX2_I2_WOFF], nq = it[CDX2_I2_NQ], part = it[CDX2_I2_PART], col0 = it[CDX2_I2_COL0];
        Cursor c;
        c.si = it[CDX2_I2_SEG]; c.tap = it[CDX2_I2_TAP]; c.cc = it[CDX2_I2_CC];
        c.s = pick(g, c.si);
        if (prof && item == 0) { asm volatile("" ::"s"(nq), "s"(c.s.ccn)); stamp(prof + 4, ptid); }
        f32x4 acc0[T][NT], acc1[T][NT];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc0[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc1[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        int roff[NT];
        lane_rows<M, NT>(g, c, col0, lane, roff);

        const f32x4* wp = reinterpret_cast<const f32x4*>(wblob + woff) + lane;
        f32x4 wr[RS][SUB];
        if (item == wave) {                                    // head of the stream was issued during the previous op
#pragma unroll
            for (int u = 0; u < RS; ++u)
#pragma unroll
                for (int h = 0; h < SUB; ++h) wr[u][h] = ring.rec[u][h];
        } else {
#pragma unroll
            for (int u = 0; u < RS; ++u)
#pragma unroll
                for (int h = 0; h < SUB; ++h) wr[u][h] = wp[(size_t)min(u * SUB + h, nq - 1) * 64];   // unconditional: vmcnt stays countable
        }
        // B-operand ring: with ONE wave per SIMD nothing else hides the ~100+ cycle ds_read latency, so the operand of chunk
        // q + BD - 1 is requested before chunk q's MFMAs issue (a chunk is only 32 cycles of matrix pipe in the 4x4 mode).
        f32x4 bq[BD][T][NT];
        auto advance = [&]() {
            if (++c.cc == c.s.ccn) {
                c.cc = 0;
                if (++c.tap == c.s.taps) {
                    c.tap = 0;
                    c.s = pick(g, ++c.si);
                }
                lane_rows<M, NT>(g, c, col0, lane, roff);
            }
        };
#pragma unroll
        for (int j = 0; j < BD - 1; ++j) {
            if (j < nq) {
                if (j > 0) advance();
                fetch_b<M, NT, T>(lds, tf, c, roff, bq[j]);
            }
        }

        if (prof && item == 0) { asm volatile("" ::"v"(wr[0][0][0]), "v"(bq[0][0][0][0])); stamp(prof + 5, ptid); }

        // chunk at ring slot u (record index == u mod PF, PF % BD == 0 -> operand slot u % BD is static after unrolling)
        auto chunk = [&](const f32x4 a, const int u, bool more) {
            if (more) {
                advance();
                fetch_b<M, NT, T>(lds, tf, c, roff, bq[(u + BD - 1) % BD]);
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc0[t][nt] = M::mfma(a[0], bq[u % BD][t][nt][0], acc0[t][nt]);
                    acc1[t][nt] = M::mfma(a[1], bq[u % BD][t][nt][1], acc1[t][nt]);
                    acc0[t][nt] = M::mfma(a[2], bq[u % BD][t][nt][2], acc0[t][nt]);
                    acc1[t][nt] = M::mfma(a[3], bq[u % BD][t][nt][3], acc1[t][nt]);
                }
        };

        // The refill of a record is issued one chunk LATE (after the next chunk's MFMAs): the scheduler may hoist a load
        // over the MFMAs of its own basic block, and a refill overlapping the last reads of the value it replaces makes the
        // register allocator double-buffer the whole ring (16 v_mov_b64 + a vmcnt(0) drain per revolution, seen in the ISA).
        // Lagged by a chunk, the old value is dead a full basic block earlier and every slot keeps its registers.
        auto refill = [&](const int slot, int q) {            // record q -> ring slot `slot` (static after unrolling)
            wr[slot / SUB][slot % SUB] = wp[(size_t)q * 64];
        };
        // steady state: every slot is refilled unconditionally (except the very first lagged one) -> counted vmcnt waits
        // (16x16 layers are short -- tens of records -- and run entirely in the drain loop: the instruction cache is 64 KiB)
        int qi = 0;
        const int n_main = M::KSTEP == 4 ? (nq / PF - 1) * PF : 0;
        for (; qi < n_main; qi += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                chunk(wr[u / SUB][u % SUB], u, true);
                if (u > 0) refill(u - 1, qi + u - 1 + PF);
                else if (qi > 0) refill(PF - 1, qi - 1 + PF);
            }
        }
        // drain: the last (up to 2*PF - 1) records, refilling only while records remain
        for (; qi < nq; qi += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (qi + u < nq) chunk(wr[u / SUB][u % SUB], u, qi + u + BD - 1 < nq);
                const int q = qi + u - 1 + PF;                 // lagged refill of the previous position
                if (q >= PF && q < nq) refill((u + PF - 1) % PF, q);
            }
        }
        if (prof && item == 0) { asm volatile("" ::"v"(acc0[0][0][0]), "v"(acc1[0][0][0])); stamp(prof + 6, ptid); }
        // D fragment: 4 consecutive rows (channels) of one column -> stage[k slice][position][row tile + rows]
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = col0 + nt * M::COLS + M::col(lane);
                if (n < g.l_out) {
                    const f32x4 d = acc0[t][nt] + acc1[t][nt];
                    *reinterpret_cast<f32x4*>(lds + t * tf + g.stage + part + n * g.sstride + M::drow(lane)) = d;
                }
            }
    }
}

struct EpiParams { f32x4 bi, ga, be, em; };

// Epilogue of one op for one trajectory region `tl`.  Thread -> (group g = tid / 32, float4 item li + 32 k): 4 consecutive
// channels c..c+3 of position pos0 + k * pstep.  NK = items per lane (compile-time so the values stay in registers).
template <int NK>
__device__ __forceinline__ void epilogue(float* __restrict__ tl, const EpiParams& P, int flags, int c, int pos0, int pstep,
                                         int li, int nv, int l_out, int c_out, int sstride, int stage, int ksplit, int dst,
                                         int dstride, int res, int rstride, float inv_cnt, int lane) {
    f32x4 v[NK];
    bool ok[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        ok[k] = li + 32 * k < nv;
        const int pos = ok[k] ? pos0 + k * pstep : 0;
        f32x4 acc = P.bi;
        for (int ks = 0; ks < ksplit; ++ks) acc += *reinterpret_cast<const f32x4*>(tl + stage + (ks * l_out + pos) * sstride + c);
        v[k] = acc;
    }
    if (flags & CDX2_F2_GN) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) s += ok[k] ? (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]) : 0.f;
        const float mean = half_sum(s, lane) * inv_cnt;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const f32x4 d = v[k] - mean;
            s2 += ok[k] ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f;
        }
        const float rstd = __builtin_amdgcn_rsqf(half_sum(s2, lane) * inv_cnt + CDX_GN_EPS);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const f32x4 y = (v[k] - mean) * rstd * P.ga + P.be;
            v[k] = (f32x4){mish2(y[0]), mish2(y[1]), mish2(y[2]), mish2(y[3])};
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (!ok[k]) continue;
        const int pos = pos0 + k * pstep;
        f32x4 y = v[k];
        if (flags & CDX2_F2_EMB) y += P.em;
        if (flags & CDX2_F2_RES) y += *reinterpret_cast<const f32x4*>(tl + res + pos * rstride + c);
        float* o = tl + dst + pos * dstride + c;
        if (c + 3 < c_out) {
            *reinterpret_cast<f32x4*>(o) = y;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < c_out) o[j] = y[j];
        }
    }
}


// Issue the first PF records of this wave's first item of op `d` (descriptor pointer, wave-uniform).
__device__ __forceinline__ void prefetch_op(const cint* d, const cint* ops, const float* __restrict__ wblob,
                                            int lane, int wave, Ring& ring) {
    if (wave < d[CDX2_W2_NITEMS]) {
        const cint* it = ops + d[CDX2_W2_ITEMS] + wave * CDX2_ITEM_WORDS;
        const int nq = it[CDX2_I2_NQ];
        const f32x4* wp = reinterpret_cast<const f32x4*>(wblob + it[CDX2_I2_WOFF]) + lane;
#pragma unroll
        for (int u = 0; u < RS; ++u)
#pragma unroll
            for (int h = 0; h < SUB; ++h) ring.rec[u][h] = wp[(size_t)min(u * SUB + h, nq - 1) * 64];
    }
}

template <int T>
__device__ __forceinline__ void run_op(const cdx_unet2_launch& L, const cint* ops, const cint* d, const cint* dn,
                                       const float* __restrict__ emb_row, float* __restrict__ lds, int tid, Ring& pre,
                                       unsigned long long* prof) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tf = L.traj_floats;
    const int flags = d[CDX2_W2_FLAGS], c_out = d[CDX2_W2_COUT], l_out = d[CDX2_W2_LOUT], coutp = d[CDX2_W2_COUTP];
    Geom g;
    g.l_out = l_out; g.l_in = d[CDX2_W2_LIN]; g.cstride = d[CDX2_W2_CSTRIDE]; g.transposed = d[CDX2_W2_TRANSPOSED];
    g.zrow = L.zrow_off;
    g.segs = d + CDX2_W2_SEG0;
    g.sstride = d[CDX2_W2_SSTRIDE]; g.stage = L.stage_off;

    // epilogue geometry + per-channel parameters: issued now, consumed after the barrier (latency hides behind the K loop)
    const int shift = d[CDX2_W2_CG4_SHIFT];
    const int grp = tid >> 5, li = tid & 31;
    const int c = grp * (coutp >> 3) + 4 * (li & ((1 << shift) - 1));
    const int pos0 = li >> shift, pstep = 32 >> shift;
    const int nv = (coutp >> 5) * l_out;
    EpiParams P;
    P.bi = *reinterpret_cast<const f32x4*>(L.wblob + d[CDX2_W2_BOFF] + c);
    P.ga = P.be = P.em = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (flags & CDX2_F2_GN) {
        P.ga = *reinterpret_cast<const f32x4*>(L.wblob + d[CDX2_W2_GAMMA] + c);
        P.be = *reinterpret_cast<const f32x4*>(L.wblob + d[CDX2_W2_BETA] + c);
    }
    if (flags & CDX2_F2_EMB) P.em = *reinterpret_cast<const f32x4*>(emb_row + d[CDX2_W2_EMB] + c);

    // K loop -> staged partial tiles
    const cint* items = ops + d[CDX2_W2_ITEMS];
    const int n_items = d[CDX2_W2_NITEMS];
    if (d[CDX2_W2_MODE] == CDX_MODE_4X4) {
        if (d[CDX2_W2_NT] == 1) conv_kloop<M4, 1, T>(g, L.wblob, items, n_items, lds, tf, lane, wave, pre, prof);
        else conv_kloop<M4, 2, T>(g, L.wblob, items, n_items, lds, tf, lane, wave, pre, prof);
    } else {
        conv_kloop<M16, 1, T>(g, L.wblob, items, n_items, lds, tf, lane, wave, pre, prof);
    }
    // head of the next op's weight stream: flies through the barrier and the epilogue
    stamp(prof ? prof + 7 : nullptr, tid);
    prefetch_op(dn, ops, L.wblob, lane, wave, pre);
    stamp(prof ? prof + 1 : nullptr, tid);
    __syncthreads();
    stamp(prof ? prof + 2 : nullptr, tid);

    const int ksplit = d[CDX2_W2_KSPLIT];
    const int dst = d[CDX2_W2_DST], dstride = d[CDX2_W2_DST_STRIDE], res = d[CDX2_W2_RES], rstride = d[CDX2_W2_RES_STRIDE];
    const float inv_cnt = __int_as_float(d[CDX2_W2_INV_CNT]);
    const int nk = d[CDX2_W2_NK];
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        float* tl = lds + t * tf;
        if (nk == 1)
            epilogue<1>(tl, P, flags, c, pos0, pstep, li, nv, l_out, c_out, g.sstride, g.stage, ksplit, dst, dstride, res, rstride,
                        inv_cnt, lane);
        else if (nk == 2)
            epilogue<2>(tl, P, flags, c, pos0, pstep, li, nv, l_out, c_out, g.sstride, g.stage, ksplit, dst, dstride, res, rstride,
                        inv_cnt, lane);
        else
            epilogue<CDX2_MAX_NK2>(tl, P, flags, c, pos0, pstep, li, nv, l_out, c_out, g.sstride, g.stage, ksplit, dst, dstride, res,
                                   rstride, inv_cnt, lane);
    }
    __syncthreads();
    stamp(prof ? prof + 3 : nullptr, tid);
}

template <int T>
__global__ __launch_bounds__(THREADS) void cdx_unet2_kernel(const cdx_unet2_launch L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = L.horizon, D = L.dim, HD = H * D, tf = L.traj_floats;
    const int b0 = blockIdx.x * T;
    unsigned long long* lprof = reinterpret_cast<unsigned long long*>(lds + T * tf);
    const bool profiling = L.prof != nullptr && blockIdx.x == 0;
    if (profiling) stamp(lprof + (size_t)L.n_ops * 8, tid);

    // weight stream of op 0 first: it flies while the state is set up
    const cint* ops = as_const(L.ops);
    Ring pre;
    prefetch_op(ops, ops, L.wblob, lane, wave, pre);

    // ---- clear the workgroup's LDS once (zero rows, pad channels of the state slots), then load x_T ----
    for (int i = tid * 4; i < T * tf; i += THREADS * 4)
        *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (b0 + t >= L.batch) break;
        const size_t xbase = (size_t)(b0 + t) * HD;
        for (int e = tid; e < HD; e += THREADS) {
            const int n = e / D, c = e - n * D;
            lds[t * tf + L.x_off + n * L.x_stride + c] = L.x_in[xbase + e];
        }
    }
    __syncthreads();

    const int n_iter = L.n_steps > 0 ? L.n_steps : 1;
    for (int step = 0; step < n_iter; ++step) {
        const float* __restrict__ emb_row = L.emb + (size_t)step * L.emb_ld;
        const cint* d = ops;
        for (int oi = 0; oi < L.n_ops; ++oi) {
            const cint* dn = (oi + 1 < L.n_ops) ? d + CDX2_OP_WORDS : ops;     // last op prefetches op 0 of the next step
            // (profile the SECOND forward when there is one: instruction / scalar caches warm, like every later step)
            unsigned long long* pslot = (profiling && step == (L.n_steps > 1 ? 1 : 0)) ? lprof + (size_t)oi * 8 : nullptr;
            stamp(pslot, tid);
            run_op<T>(L, ops, d, dn, emb_row, lds, tid, pre, pslot);
            d = dn;
        }
        if (L.n_steps == 0) break;
        // ---- clip, eps/x0 conversion, solver update, fix-mask blend on the LDS-resident state (kinds 0-4) ----
        const cdx_step st = L.steps[step];
        const float al = st.alpha, sg = st.sigma;
        const float k0 = st.k[0], k1 = st.k[1], k2 = st.k[2], k3 = st.k[3], k4 = st.k[4];
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if (b0 + t >= L.batch) break;
            const int b = b0 + t;
            const size_t xbase = (size_t)b * HD;
            float* tl = lds + t * tf;
            for (int e = tid; e < HD; e += THREADS) {
                const int n = e / D, c = e - n * D;
                const int xo = L.x_off + n * L.x_stride + c;
                const float x = tl[xo];
                float p = tl[L.pred_off + n * L.pred_stride + c];
                if (L.predict_noise) {
                    if (L.x_max) p = fmaxf(p, (x - al * L.x_max[e]) / sg);
                    if (L.x_min) p = fminf(p, (x - al * L.x_min[e]) / sg);
                } else {
                    if (L.x_min) p = fmaxf(p, L.x_min[e]);
                    if (L.x_max) p = fminf(p, L.x_max[e]);
                }
                float eps, xth, xn;
                if (L.predict_noise) {
                    eps = p; xth = (x - sg * p) / al;
                } else {
                    xth = p; eps = (x - al * p) / sg;
                }
                if (st.kind >= 3) {
                    // legacy DDPM class (reference diffusion/ddpm.py:153-164, 230-241): fix-mask on the prediction
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
                    if (st.flags & CDX_STEP_MASK_PRED) {     // legacy DPMSolver (dpmsolver.py:257-264)
                        const float m = L.fix_mask ? L.fix_mask[e] : 0.f;
                        eps = eps * (1.0f - m);
                        xth = xth * (1.0f - m) + x * m;
                    }
                    float v = (st.vsel & 1) ? xth : eps;
                    if (st.vsel == 2) v = k3 * xth - k4 * tl[L.prev_off + e];
                    if (st.vsel == 3) v = k3 * eps - k4 * tl[L.prev_off + e];
                    xn = k0 * x - k1 * v;
                    if (st.noise_idx >= 0) xn += k2 * L.noise[((size_t)st.noise_idx * L.batch + b) * HD + e];
                }
                if (L.fix_mask) {
                    const float m = L.fix_mask[e];
                    xn = xn * (1.0f - m) + L.prior[xbase + e] * m;
                }
                if (st.push) tl[L.prev_off + e] = st.push == 2 ? eps : xth;
                tl[xo] = xn;
            }
        }
        __syncthreads();
    }
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (b0 + t >= L.batch) break;
        const size_t xbase = (size_t)(b0 + t) * HD;
        const int off = L.n_steps == 0 ? L.pred_off : L.x_off, str = L.n_steps == 0 ? L.pred_stride : L.x_stride;
        for (int e = tid; e < HD; e += THREADS) {
            const int n = e / D, c = e - n * D;
            L.x_out[xbase + e] = lds[t * tf + off + n * str + c];
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
        m[o] = mish2(acc);
    }
    __syncthreads();
    for (int o = tid; o < A.n_emb; o += 256) {
        float acc = A.wblob[A.b3 + o];
        for (int i = 0; i < A.md; ++i) acc = fmaf(A.wblob[A.w3 + (size_t)i * A.n_emb + o], m[i], acc);
        A.out[(size_t)r * A.n_emb + o] = acc;
    }
}

}  // namespace

extern "C" {

int cdx_unet2_embtab(const cdx_unet2_embtab_args* A, void* hip_stream) {
    cdx_set_err("");
    if (!A || !A->wblob || !A->temb || !A->out) { cdx_set_err("null pointer in embtab args"); return CDX_EINVAL; }
    if (A->n_rows == 0) return CDX_OK;
    if (A->n_rows < 0 || A->emb_dim <= 0 || A->hidden <= 0 || A->md <= 0 || A->n_emb <= 0) { cdx_set_err("non-positive size"); return CDX_EINVAL; }
    const size_t sh = (size_t)(A->emb_dim + A->hidden + A->md) * sizeof(float);
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
    if (L->n_ops <= 0 || L->batch < 0 || L->horizon <= 0 || L->dim <= 0 || L->traj_floats <= 0) { cdx_set_err("non-positive size"); return CDX_EINVAL; }
    if (L->traj_per_wg != 1 && L->traj_per_wg != 2) { cdx_set_err("traj_per_wg must be 1 or 2"); return CDX_EINVAL; }
    if (L->n_steps > 0 && !L->steps) { cdx_set_err("steps == NULL with n_steps > 0"); return CDX_EINVAL; }
    if (L->n_steps < 0) { cdx_set_err("negative n_steps"); return CDX_EINVAL; }
    if (L->fix_mask && !L->prior) { cdx_set_err("fix_mask given without prior"); return CDX_EINVAL; }
    if ((L->traj_floats | L->zrow_off | L->x_off | L->x_stride | L->pred_off | L->pred_stride | L->prev_off | L->stage_off | L->emb_ld) & 3) {
        cdx_set_err("LDS offsets/strides and emb_ld must be multiples of 4 floats"); return CDX_EINVAL;
    }
    size_t lds_bytes = (size_t)L->traj_floats * L->traj_per_wg * sizeof(float);
    if (L->prof) lds_bytes += (size_t)(L->n_ops * 8 + 2) * sizeof(unsigned long long);
    if (lds_bytes > 160u * 1024u) { cdx_set_err("program needs more than 160 KiB of LDS"); return CDX_ELDS; }
    auto kern = L->traj_per_wg == 2 ? cdx_unet2_kernel<2> : cdx_unet2_kernel<1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    const int grid = (L->batch + L->traj_per_wg - 1) / L->traj_per_wg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds_bytes, reinterpret_cast<hipStream_t>(hip_stream), *L);
    e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

}  // extern "C"
