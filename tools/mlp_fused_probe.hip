// Probe for DESIGN.md section 7, open item 1: fc1 -> GELU -> fc2 of a DiT block as ONE kernel, the 4x hidden never leaving the CU.
//
//   Y[m][:] = b2 + W2 . gelu_tanh(b1 + W1 . X[m][:])        X: (M, 320), W1: (1280, 320), W2: (320, 1280)  (torch Linear layouts)
//
// Shape of the kernel (the plan written down in DESIGN.md): a workgroup owns 64 rows and all 320 output columns; four wave64, each
// 32 rows x 160 columns of Y (5 v_mfma_f32_32x32x2_f32 tiles = 80 accumulator VGPRs) and 32 rows x 64 columns of the current hidden
// chunk (2 tiles).  The hidden dimension goes by in 10 chunks of 128: phase 1 streams 16-wide K slabs of X and W1 through LDS
// ([k][row] tiles, two stages, one barrier per slab -- the loop of csrc/cdx_gemm.hip), GELU is applied on the accumulators and the
// chunk is parked in LDS as the A operand of phase 2, which streams 8-wide K slabs of W2.  79.6 KB of LDS: two workgroups per CU.
// Weights are read from a SLAB-MAJOR repack made once on the host (they are constants of the sampling loop): W1p[chunk][slab][k][hidden
// unit], W2p[chunk][slab][k][output column] -- a slab is one contiguous block, fetched with fully coalesced float4 loads and written to
// LDS as float4 without a transposition (the first version of this probe read torch-layout rows, one 16-byte piece of a different row
// per lane: 55.0 % of peak).  The first X / W1 slab of the next chunk is requested during the last W2 slab of the current one.
// Measured state at the end of round 3 (profiles/r03_mlp_fused_probe.txt): results exact to 1.3e-6 of a float64 reference; 0.62-0.65 ms per
// call = 53-55 % of peak, phase 2 at 57 % of its MFMA floor -- and the ISA of THAT build shows why: the W2 staging registers were a guarded
// float4[3] that the compiler kept in scratch, every global load followed by s_waitcnt vmcnt(0) and a scratch store.  This file carries the
// fix (named registers, unconditional clamped loads); it compiles to 0 scratch bytes and has NOT been timed yet.  Two compile-time knobs prepare
// the first measurements of the next round (tools/gpu_mlp_fused_probe.sh builds and runs all four): PROBE_BK2 (W2 slab of 8 or 16 k) and
// PROBE_UPFRONT (LDS operands of a slab requested before its first MFMA).  Every build checks itself against the float64 reference.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/mlp_fused_probe tools/mlp_fused_probe.hip && tools/_bin/mlp_fused_probe
// prints: max relative error against a float64 host reference on sampled rows, time per call, TFLOP/s and the fraction of the
// 157.3 TFLOP/s fp32-MFMA peak -- to be read next to the two-GEMM figures of profiles/ (fc1 62-67 %, fc2 66.5 % of peak).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef PROBE_BK2
#define PROBE_BK2 8          // K slab of phase 2: 8 (79.6 KB of LDS, two workgroups per CU) or 16 (100 KB, one workgroup per CU, half the barriers)
#endif
#ifndef PROBE_UPFRONT
#define PROBE_UPFRONT 0      // 1: all LDS operands of a slab are requested before its first MFMA
#endif
constexpr int D = 320, HID = 1280, BM = 64, CH = 128, BK = 16, BK2 = PROBE_BK2, THREADS = 256;
constexpr int LDX = BM + 4, LDW1 = CH + 4, LDH = BM + 1, LDW2 = D + 4;
constexpr int SMEM_FLOATS = 2 * BK * LDX + 2 * BK * LDW1 + CH * LDH + 2 * BK2 * LDW2;

__device__ __forceinline__ float gelu_tanh(float x) {       // 0.5 x (1 + tanh u) == x * sigmoid(2u)   (csrc/cdx_gemm.hip: gm_act)
    const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
}

__global__ __launch_bounds__(THREADS, 2) void mlp_fused_kernel(const float* __restrict__ X, const float* __restrict__ W1,
                                                                const float* __restrict__ b1, const float* __restrict__ W2,
                                                                const float* __restrict__ b2, float* __restrict__ Y, int M,
                                                                unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*Xs)[LDX] = reinterpret_cast<float (*)[LDX]>(smem);                                        // [2 BK][LDX]   X slabs,  [k][row]
    float (*W1s)[LDW1] = reinterpret_cast<float (*)[LDW1]>(smem + 2 * BK * LDX);                       // [2 BK][LDW1]  W1 slabs, [k][hidden unit]
    float (*Hs)[LDH] = reinterpret_cast<float (*)[LDH]>(smem + 2 * BK * LDX + 2 * BK * LDW1);          // [CH][LDH]     gelu chunk, [hidden unit][row]
    float (*W2s)[LDW2] = reinterpret_cast<float (*)[LDW2]>(smem + 2 * BK * LDX + 2 * BK * LDW1 + CH * LDH);   // [2 BK2][LDW2] W2 slabs, [k][out col]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bm = blockIdx.x * BM;
    const int rh = wave >> 1, chh = wave & 1;           // this wave: rows rh*32.., hidden columns chh*64.. (phase 1), output columns chh*160.. (phase 2)
    const int lr = lane & 31, lk = lane >> 5;

    // staging roles
    const int xr = tid & 63, xq = tid >> 6;              // X slab: row xr, k quad xq (k = 4 xq)
    const float* xp = X + (size_t)min(bm + xr, M - 1) * D + xq * 4;
    constexpr int N4_2 = BK2 * D / 4, R2 = (N4_2 + THREADS - 1) / THREADS;      // float4 of a W2 slab, rounds of 256 threads
    float4 rx, rw0, rw1, r2a, r2b, r2c, r2d, r2e;      // (named registers: a float4[R2] captured by the staging lambdas stays in scratch)
    static_assert(R2 == 3 || R2 == 5, "W2 slab of 8 or 16 k");

    unsigned long long t_p1 = 0, t_gelu = 0, t_p2 = 0;          // cycles per phase (workgroups 0 and 300 report)
    f32x16 acc2[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

#pragma unroll 1
    for (int c = 0; c < HID / CH; ++c) {
        // W1p slab (c, t): 16 k x 128 hidden units = 512 contiguous float4, two per thread; LDS row k, columns 4 n4 .. 4 n4 + 3
        auto fetch1 = [&](int cc, int t) {
            rx = *reinterpret_cast<const float4*>(xp + t * BK);
            const float4* src = reinterpret_cast<const float4*>(W1 + ((size_t)cc * (D / BK) + t) * (BK * CH));
            rw0 = src[tid];
            rw1 = src[tid + THREADS];
        };
        auto stage1 = [&](int buf) {
            float (*Xd)[LDX] = Xs + buf * BK;
            float (*Wd)[LDW1] = W1s + buf * BK;
            const int kx = xq * 4;
            Xd[kx + 0][xr] = rx.x; Xd[kx + 1][xr] = rx.y; Xd[kx + 2][xr] = rx.z; Xd[kx + 3][xr] = rx.w;
            *reinterpret_cast<float4*>(&Wd[tid >> 5][(tid & 31) * 4]) = rw0;
            *reinterpret_cast<float4*>(&Wd[8 + (tid >> 5)][(tid & 31) * 4]) = rw1;
        };
        // W2p slab (c, u): 8 k x 320 output columns = 640 contiguous float4, three rounds of 256 threads (the last one half empty)
        auto fetch2 = [&](int u) {                       // (unconditional clamped loads: guarded ones left the registers in scratch)
            const float4* src = reinterpret_cast<const float4*>(W2 + ((size_t)c * (CH / BK2) + u) * (BK2 * D));
            r2a = src[tid];
            r2b = src[tid + THREADS];
            r2c = src[min(tid + 2 * THREADS, N4_2 - 1)];
            if (R2 == 5) { r2d = src[min(tid + 3 * THREADS, N4_2 - 1)]; r2e = src[min(tid + 4 * THREADS, N4_2 - 1)]; }
        };
        auto stage2 = [&](int buf) {
            float (*Wd)[LDW2] = W2s + buf * BK2;
            constexpr int Q = D / 4;                     // float4 per k row
            auto put = [&](int idx, const float4& v) { if (idx < N4_2) *reinterpret_cast<float4*>(&Wd[idx / Q][(idx % Q) * 4]) = v; };
            put(tid, r2a);
            put(tid + THREADS, r2b);
            put(tid + 2 * THREADS, r2c);
            if (R2 == 5) { put(tid + 3 * THREADS, r2d); put(tid + 4 * THREADS, r2e); }
        };

        // ---------------- phase 1: chunk = X[64 x 320] . W1[chunk]^T ----------------
        f32x16 acc1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
        constexpr int NK1 = D / BK;                      // 20 slabs
        const unsigned long long s0 = clock64();
        if (c == 0) fetch1(0, 0);                        // (later chunks: requested during the previous chunk's last W2 slab)
        stage1(0);
        fetch1(c, 1);
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < NK1; ++t) {
            const float (*Xc)[LDX] = Xs + (t & 1) * BK;
            const float (*Wc)[LDW1] = W1s + (t & 1) * BK;
#if PROBE_UPFRONT
            float av[BK / 2], bv0[BK / 2], bv1[BK / 2];
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp) {
                av[kp] = Xc[2 * kp + lk][rh * 32 + lr];
                bv0[kp] = Wc[2 * kp + lk][chh * 64 + lr];
                bv1[kp] = Wc[2 * kp + lk][chh * 64 + 32 + lr];
            }
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp) {
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp], bv0[kp], acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp], bv1[kp], acc1[1], 0, 0, 0);
                if (kp == BK / 4 - 1 && t + 1 < NK1) {
                    stage1((t + 1) & 1);
                    if (t + 2 < NK1) fetch1(c, t + 2);
                    else fetch2(0);
                }
            }
#else
            float av = Xc[lk][rh * 32 + lr], bv0 = Wc[lk][chh * 64 + lr], bv1 = Wc[lk][chh * 64 + 32 + lr];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float na = 0.f, nb0 = 0.f, nb1 = 0.f;
                if (kk + 2 < BK) { na = Xc[kk + 2 + lk][rh * 32 + lr]; nb0 = Wc[kk + 2 + lk][chh * 64 + lr]; nb1 = Wc[kk + 2 + lk][chh * 64 + 32 + lr]; }
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv0, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv1, acc1[1], 0, 0, 0);
                if (kk == BK / 2 - 2 && t + 1 < NK1) {   // mid-slab: park slab t + 1 in the other stage, request slab t + 2
                    stage1((t + 1) & 1);
                    if (t + 2 < NK1) fetch1(c, t + 2);
                    else fetch2(0);                       // (the last phase-1 slab of the chunk: first W2 slab instead)
                }
                av = na; bv0 = nb0; bv1 = nb1;
            }
#endif
            __syncthreads();
        }
        const unsigned long long s1 = clock64();
        // bias + GELU on the accumulators, parked as the A operand of phase 2: Hs[hidden unit][row]
        // (D fragment of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = chh * 64 + 32 * j + lr;
            const float bias = b1[c * CH + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                Hs[col][m] = gelu_tanh(acc1[j][r] + bias);
            }
        }
        // ---------------- phase 2: Y += chunk[64 x 128] . W2[:, chunk]^T ----------------
        constexpr int NK2 = CH / BK2;                    // 16 slabs
        stage2(0);
        fetch2(1);
        __syncthreads();
        const unsigned long long s2 = clock64();
#pragma unroll 1
        for (int u = 0; u < NK2; ++u) {
            const float (*Wc)[LDW2] = W2s + (u & 1) * BK2;
            const float (*Hc)[LDH] = Hs + u * BK2;
#if PROBE_UPFRONT
            float av[BK2 / 2], bv[BK2 / 2][5];
#pragma unroll
            for (int kp = 0; kp < BK2 / 2; ++kp) {
                av[kp] = Hc[2 * kp + lk][rh * 32 + lr];
#pragma unroll
                for (int j = 0; j < 5; ++j) bv[kp][j] = Wc[2 * kp + lk][chh * 160 + 32 * j + lr];
            }
#pragma unroll
            for (int kp = 0; kp < BK2 / 2; ++kp) {
#pragma unroll
                for (int j = 0; j < 5; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp], bv[kp][j], acc2[j], 0, 0, 0);
                if (kp == BK2 / 4 - 1) {
                    if (u + 1 < NK2) {
                        stage2((u + 1) & 1);
                        if (u + 2 < NK2) fetch2(u + 2);
                    } else if (c + 1 < HID / CH) fetch1(c + 1, 0);
                }
            }
#else
#pragma unroll
            for (int kk = 0; kk < BK2; kk += 2) {
                const float av = Hc[kk + lk][rh * 32 + lr];
                float bv[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) bv[j] = Wc[kk + lk][chh * 160 + 32 * j + lr];
#pragma unroll
                for (int j = 0; j < 5; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc2[j], 0, 0, 0);
                if (kk == BK2 / 2 - 2) {
                    if (u + 1 < NK2) {
                        stage2((u + 1) & 1);
                        if (u + 2 < NK2) fetch2(u + 2);
                    } else if (c + 1 < HID / CH) fetch1(c + 1, 0);
                }
            }
#endif
            __syncthreads();
        }
        const unsigned long long s3 = clock64();
        t_p1 += s1 - s0; t_gelu += s2 - s1; t_p2 += s3 - s2;
    }
    if (stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 300)) {
        unsigned long long* o = stamps + (blockIdx.x ? 3 : 0);
        o[0] = t_p1; o[1] = t_gelu; o[2] = t_p2;
    }
    // ---------------- epilogue: + b2, straight from the D fragments (32 consecutive columns per row and register) ----------------
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int n = chh * 160 + 32 * j + lr;
        const float bias = b2[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = bm + rh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < M) Y[(size_t)m * D + n] = acc2[j][r] + bias;
        }
    }
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768;      // config 4 at its 512-shard: 512 x 64 tokens
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s  CUs %d   M = %d rows, d = %d, hidden = %d   LDS per workgroup %.1f KB   [W2 slab k = %d, operands up front = %d]\n", prop.name,
           prop.multiProcessorCount, M, D, HID, SMEM_FLOATS * 4 / 1024.0, BK2, PROBE_UPFRONT);
    unsigned seed = 12345u;
    std::vector<float> hX((size_t)M * D), hW1((size_t)HID * D), hW2((size_t)D * HID), hb1(HID), hb2(D);
    for (auto& v : hX) v = frand(seed);
    for (auto& v : hW1) v = frand(seed) * 0.08f;
    for (auto& v : hW2) v = frand(seed) * 0.04f;
    for (auto& v : hb1) v = frand(seed) * 0.1f;
    for (auto& v : hb2) v = frand(seed) * 0.1f;
    // slab-major repack (once per weight version in a product path): W1p[c][t][k][j] = W1[c*128 + j][t*16 + k],
    //                                                               W2p[c][u][k][n] = W2[n][c*128 + u*8 + k]
    std::vector<float> pW1(hW1.size()), pW2(hW2.size());
    for (int c = 0; c < HID / CH; ++c)
        for (int t = 0; t < D / BK; ++t)
            for (int k = 0; k < BK; ++k)
                for (int j = 0; j < CH; ++j)
                    pW1[(((size_t)c * (D / BK) + t) * BK + k) * CH + j] = hW1[(size_t)(c * CH + j) * D + t * BK + k];
    for (int c = 0; c < HID / CH; ++c)
        for (int u = 0; u < CH / BK2; ++u)
            for (int k = 0; k < BK2; ++k)
                for (int n = 0; n < D; ++n)
                    pW2[(((size_t)c * (CH / BK2) + u) * BK2 + k) * D + n] = hW2[(size_t)n * HID + c * CH + u * BK2 + k];
    float *X, *W1, *W2, *b1, *b2, *Y;
    CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&W1, hW1.size() * 4)); CK(hipMalloc(&W2, hW2.size() * 4));
    CK(hipMalloc(&b1, hb1.size() * 4)); CK(hipMalloc(&b2, hb2.size() * 4)); CK(hipMalloc(&Y, (size_t)M * D * 4));
    CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W1, pW1.data(), pW1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W2, pW2.data(), pW2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b1, hb1.data(), hb1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b2, hb2.data(), hb2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Y, 0xff, (size_t)M * D * 4));
    const size_t lds_bytes = SMEM_FLOATS * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const int grid = (M + BM - 1) / BM;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, 6 * sizeof(unsigned long long)));
    CK(hipMemset(stamps, 0, 6 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(mlp_fused_kernel, dim3(grid), dim3(THREADS), lds_bytes, 0, X, W1, b1, W2, b2, Y, M, stamps);
    CK(hipDeviceSynchronize());
    // float64 reference on sampled rows
    std::vector<float> hY((size_t)M * D);
    CK(hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, ymax = 0;
    int nan_count = 0;
    std::vector<double> h(HID);
    for (int s = 0; s < 24; ++s) {
        const int m = s < 2 ? (s == 0 ? 0 : M - 1) : (int)(((long long)s * 2654435761u) % M);
        for (int j = 0; j < HID; ++j) {
            double a = hb1[j];
            for (int k = 0; k < D; ++k) a += (double)hW1[(size_t)j * D + k] * hX[(size_t)m * D + k];
            h[j] = 0.5 * a * (1.0 + std::tanh(0.7978845608028654 * (a + 0.044715 * a * a * a)));
        }
        for (int n = 0; n < D; ++n) {
            double y = hb2[n];
            for (int j = 0; j < HID; ++j) y += (double)hW2[(size_t)n * HID + j] * h[j];
            const double got = hY[(size_t)m * D + n];
            if (!(got == got)) { ++nan_count; continue; }
            worst = std::fmax(worst, std::fabs(got - y));
            ymax = std::fmax(ymax, std::fabs(y));
        }
    }
    printf("check on 24 rows x %d columns: max |err| %.3e against max |y| %.3f  (relative %.2e)%s\n", D, worst, ymax, worst / ymax,
           nan_count ? "   [NaN / unwritten outputs]" : "");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mlp_fused_kernel, dim3(grid), dim3(THREADS), lds_bytes, 0, X, W1, b1, W2, b2, Y, M, stamps);
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mlp_fused_kernel, dim3(grid), dim3(THREADS), lds_bytes, 0, X, W1, b1, W2, b2, Y, M, stamps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = 2.0 * M * (double)D * HID * 2.0;
    printf("fused fc1 -> GELU -> fc2: %.3f ms per call, %.1f TFLOP/s = %.1f %% of the 157.3 TFLOP/s fp32-MFMA peak (%d workgroups)\n", ms,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3 * 100.0, grid);
    unsigned long long hs[6];
    CK(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
    for (int b = 0; b < 2; ++b)
        printf("workgroup %3d, cycles per call: phase 1 (X . W1^T, 2 MFMAs per k pair) %llu  gelu + park + first W2 slab %llu  phase 2 (chunk . W2^T, 5 MFMAs per k pair) %llu"
               "   [MFMA-only floor: 204800 per phase]\n", b ? 300 : 0, hs[3 * b], hs[3 * b + 1], hs[3 * b + 2]);
    return 0;
}
