// Probe: can the head of the NEXT op's weight stream be brought into LDS while the current op's epilogue runs?
//
// DESIGN.md section 7 (open item: B = 256 of config 2): ~30 % of a forward is per-op latency during which the L2 -> CU path idles
// (decode, barriers, epilogue ~1.7 k cycles per op on waves 0-3 while waves 4-7 only rewrite halo rows).  The one idea left is to let
// the idle waves issue direct-to-LDS loads (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs) for the first 32-96 KB of the
// next op's records into the spare LDS, and to feed the K loop's first records from there.  What that needs to be worth building:
//   (1) the DMA must run at stream rate next to an epilogue-like VALU / LDS load on the other four waves, without slowing it;
//   (2) reading the records back (ds_read_b128, one per record per wave) must be cheap next to the ~21 clk a record costs to stream.
// This probe measures both on all 256 CUs at once (every workgroup streams the SAME addresses, like the kernel: L2 / MALL hits),
// and the register-ring form of the same bytes (global_load_dwordx4 into VGPRs on 8 waves: what the kernel does today) beside it.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/lds_dma_probe tools/lds_dma_probe.hip && tools/_bin/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 512;
constexpr int DMA_FLOATS = 96 * 256;          // up to 96 KB of records
constexpr int WORK_FLOATS = 8192;             // 32 KB: the "destination slot" the epilogue-like loop works on
constexpr size_t W_FLOATS = 4u << 20;         // 16 MB of "weights" (the config-2 set is 15.9 MB)

__device__ __forceinline__ void dma_piece(const float* __restrict__ g, float* lds_wave_base) {
    // 1 KiB: lane l's 16 bytes land at lds_wave_base + 4 l floats (destination = wave-uniform base + lane x 16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same piece with M0 written in the statement that uses it (M0 is compiler-reserved; one wait state between the write and the load)
__device__ __forceinline__ void dma_piece_asm(const float* __restrict__ g, float* lds_wave_base) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

#define DMA_PIECE(g, l) do { if (use_asm) dma_piece_asm((g), (l)); else dma_piece((g), (l)); } while (0)

__device__ __forceinline__ float epi_work(float* work, int tid, int e, float acc) {
    // one "float4 item" of a GroupNorm -> Mish epilogue: read, ~20 VALU ops with transcendentals, write
    f32x4 v = *reinterpret_cast<f32x4*>(work + ((tid * 4 + e * 1024) & (WORK_FLOATS - 4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[j] * 1.0001f + acc * 1e-9f;
        const float sp = __logf(1.f + __expf(a));
        const float t = 1.f - 2.f / (1.f + __expf(2.f * sp));
        v[j] = a * t;
    }
    *reinterpret_cast<f32x4*>(work + ((tid * 4 + e * 1024 + 2048) & (WORK_FLOATS - 4))) = v;
    return acc + v[0];
}

// variant 0: epilogue-like loop alone (waves 0-3), nothing streamed
//         1: DMA alone, issued by waves 4-7 (dma_kb pieces over 4 waves), waves 0-3 idle at the barrier
//         2: both: waves 0-3 epilogue loop, waves 4-7 DMA
//         3: both, DMA issued by ALL waves (waves 0-3 one piece every other item of their loop)
//         4: register stream alone: the same bytes as global_load_dwordx4 into VGPRs on 8 waves, 8 in flight per lane
//         5: register stream on waves 4-7 next to the epilogue loop on waves 0-3
__global__ __launch_bounds__(THREADS) void probe(const float* __restrict__ w, int iters, int variant, int dma_kb, int epi_iters,
                                                 unsigned long long* cyc, float* sink, int* bad, int use_asm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dma = lds;
    float* work = lds + DMA_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < WORK_FLOATS; i += THREADS) work[i] = (float)(i & 31) * 0.01f;
    __syncthreads();
    float acc = 0.f;
    int wrong = 0;
    unsigned long long tE = 0, tK = 0;
    const int pieces = dma_kb;                       // 1 KiB each
    for (int it = 0; it < iters; ++it) {
        const size_t base = ((size_t)it * pieces * 256) % (W_FLOATS - (size_t)pieces * 256);
        __syncthreads();
        const unsigned long long t0 = clock64();
        // ---------------- phase E: epilogue-like work and / or the fetch of the next head ----------------
        if (variant == 4 || (variant == 5 && wave >= 4)) {
            const int nw = variant == 4 ? 8 : 4, w0 = variant == 4 ? wave : wave - 4;
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int p = w0; p < pieces; p += nw * 8) {
                f32x4 r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pp = p + u * nw;
                    r[u] = pp < pieces ? *reinterpret_cast<const f32x4*>(w + base + (size_t)pp * 256 + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += r[u];
            }
            acc += s[0] + s[1] + s[2] + s[3];
        }
        if ((variant == 1 || variant == 2) && wave >= 4) {
            for (int p = wave - 4; p < pieces; p += 4) DMA_PIECE(w + base + (size_t)p * 256 + lane * 4, dma + p * 256);
        }
        if (variant == 3 && wave >= 4) {
            for (int p = wave; p < pieces; p += 8) DMA_PIECE(w + base + (size_t)p * 256 + lane * 4, dma + p * 256);
        }
        if (variant != 1 && variant != 4 && wave < 4) {
            int p = wave;
            for (int e = 0; e < epi_iters; ++e) {
                acc = epi_work(work, tid, e, acc);
                if (variant == 3 && (e & 1) == 0 && p < pieces) { DMA_PIECE(w + base + (size_t)p * 256 + lane * 4, dma + p * 256); p += 8; }
            }
            if (variant == 3) for (; p < pieces; p += 8) DMA_PIECE(w + base + (size_t)p * 256 + lane * 4, dma + p * 256);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t1 = clock64();
        // ---------------- phase K: the records come back out of LDS (one ds_read_b128 per record per wave) ----------------
        if (variant == 1 || variant == 2 || variant == 3) {
            for (int p = wave; p < pieces; p += 8) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dma + p * 256 + lane * 4);
                const size_t i0 = base + (size_t)p * 256 + lane * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) wrong += (it == iters - 1) && v[j] != (float)((i0 + j) & 0xFFFF);
                acc += v[0];
            }
        }
        __syncthreads();
        const unsigned long long t2 = clock64();
        tE += t1 - t0; tK += t2 - t1;
    }
    if (wrong) atomicAdd(bad, wrong);
    if (tid == 0) { cyc[2 * blockIdx.x] = tE; cyc[2 * blockIdx.x + 1] = tK; }
    if (acc == 123.456f) sink[tid] = acc;
}

// layout check: ONE wave issues FOUR 1 KiB pieces back to back (piece p: w[256 p ...] -> lds[256 (1 + p) ...]), waits, and the workgroup
// dumps the LDS.  mode 0: the builtin in a loop; 1: the asm statement (M0 written next to its use); 2: the builtin, one piece at a time
__global__ __launch_bounds__(THREADS) void layout_kernel(const float* __restrict__ w, float* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += THREADS) lds[i] = -1.f;
    __syncthreads();
    if (wave == 5) {
        for (int p = 0; p < 4; ++p) {
            if (mode == 1) dma_piece_asm(w + p * 256 + lane * 4, lds + 256 * (1 + p));
            else dma_piece(w + p * 256 + lane * 4, lds + 256 * (1 + p));
            if (mode == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2048; i += THREADS) out[i] = lds[i];
}

static void layout(const float* w) {
    float* out;
    CK(hipMalloc(&out, 2048 * sizeof(float)));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(THREADS), 8192, 0, w, out, mode);
        CK(hipDeviceSynchronize());
        std::vector<float> h(2048);
        CK(hipMemcpy(h.data(), out, 2048 * sizeof(float), hipMemcpyDeviceToHost));
        printf("four pieces back to back, mode %d (%s):", mode, mode == 0 ? "builtin" : mode == 1 ? "asm, M0 beside the load" : "builtin, vmcnt(0) after each");
        for (int p = 0; p < 4; ++p) {
            int ok = 0, untouched = 0;
            for (int i = 0; i < 256; ++i) { ok += h[256 * (1 + p) + i] == (float)(256 * p + i); untouched += h[256 * (1 + p) + i] == -1.f; }
            printf("  piece %d: %3d / 256 right, %3d untouched, first word %g", p, ok, untouched, h[256 * (1 + p)]);
        }
        int stray = 0;
        for (int i = 0; i < 2048; ++i) stray += (i < 256 || i >= 1280) && h[i] != -1.f;
        printf("  stray words %d\n", stray);
    }
    CK(hipFree(out));
}

static void run(const float* w, int variant, int dma_kb, int epi_iters, const char* tag, int use_asm = 0) {
    const int wgs = 256, iters = 200;
    unsigned long long* cyc; float* sink; int* bad;
    CK(hipMalloc(&cyc, 2 * wgs * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, THREADS * sizeof(float)));
    CK(hipMalloc(&bad, sizeof(int)));
    CK(hipMemset(bad, 0, sizeof(int)));
    const size_t lds_bytes = (DMA_FLOATS + WORK_FLOATS) * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    for (int rep = 0; rep < 2; ++rep) {               // (first pass warms L2 / MALL and the instruction cache)
        hipLaunchKernelGGL(probe, dim3(wgs), dim3(THREADS), lds_bytes, 0, w, iters, variant, dma_kb, epi_iters, cyc, sink, bad, use_asm);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(2 * wgs);
    int hbad = 0;
    CK(hipMemcpy(h.data(), cyc, 2 * wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost));
    double e = 0, k = 0;
    for (int i = 0; i < wgs; ++i) { e += (double)h[2 * i]; k += (double)h[2 * i + 1]; }
    e /= (double)wgs * iters; k /= (double)wgs * iters;
    const double bpc = variant == 0 ? 0.0 : dma_kb * 1024.0 / e;
    printf("%-44s %s kb %3d epi %3d : phase E %7.0f cycles (%5.1f B/clk/CU if it were all stream)   read-back %6.0f cycles%s\n", tag, use_asm ? "asm" : "bi ", dma_kb, epi_iters,
           e, bpc, k, hbad ? "   [WRONG DATA]" : "");
    CK(hipFree(cyc)); CK(hipFree(sink)); CK(hipFree(bad));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s  CUs %d\n", prop.name, prop.multiProcessorCount);
    std::vector<float> hw(W_FLOATS);
    for (size_t i = 0; i < W_FLOATS; ++i) hw[i] = (float)(i & 0xFFFF);
    float* w;
    CK(hipMalloc(&w, W_FLOATS * sizeof(float)));
    CK(hipMemcpy(w, hw.data(), W_FLOATS * sizeof(float), hipMemcpyHostToDevice));
    layout(w);
    for (int epi : {2, 4, 8}) run(w, 0, 0, epi, "epilogue-like loop alone (waves 0-3)");
    for (int kb : {32, 64}) {
        run(w, 1, kb, 0, "LDS DMA alone (waves 4-7 issue)");
        run(w, 1, kb, 0, "LDS DMA alone (waves 4-7 issue)", 1);
        run(w, 4, kb, 0, "register stream alone (8 waves)");
        for (int epi : {2, 4, 8}) {
            run(w, 2, kb, epi, "epilogue (0-3) + LDS DMA (4-7)", 1);
            run(w, 3, kb, epi, "epilogue + LDS DMA issued by all waves", 1);
            run(w, 5, kb, epi, "epilogue (0-3) + register stream (4-7)");
        }
    }
    return 0;
}
