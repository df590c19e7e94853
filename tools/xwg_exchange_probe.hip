// Probe: what does one cross-workgroup exchange cost inside a launch?
//
// DESIGN.md section 6 prices a small-batch mode of the program kernel (a trajectory's stream-bound layers split over k workgroups of
// one XCD, each computing 1/k of the output channels and all-gathering a 2-4 KB tile through L2 with a flag per layer).  The price of
// that all-gather is the term the estimate hinges on; this probe measures it: k workgroups (one per CU, 512 threads, co-resident by
// construction: grid <= CUs and a 96 KiB LDS allocation) publish 4 KB / k floats each, release, raise a flag, wait for the other
// k - 1 flags, acquire, and read the others' parts -- N times in a row, timed with s_memtime on the workgroup itself.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/xwg_exchange_probe tools/xwg_exchange_probe.hip
//   tools/_bin/xwg_exchange_probe      # one line per variant: cycles (and us) per exchange
//
// Every spin is bounded (a lost flag ends the kernel with an error code instead of hanging the GPU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 512;
constexpr int TILE_FLOATS = 1024;      // 4 KB: 256 channels x 4 positions, the output of one L = 4 layer

// group g = the k workgroups {base + stride * j}: stride 8 keeps a group on ONE XCD (workgroup i runs on XCD i % 8), stride 1
// spreads it over k XCDs
// mode 0: every thread fences (agent-scope release before the barrier, acquire after the wait);
// mode 1: ONE lane fences (all stores drained by a workgroup-scope release + barrier first; one buffer_wbl2 / buffer_inv per CU);
// mode 2: no agent-scope fence at all -- valid for a group on ONE XCD (one L2): stores drained at workgroup scope, flag, and the
//         readers bypass their L1 with agent-scope relaxed atomic loads (sc1)
__global__ __launch_bounds__(THREADS) void exchange_kernel(float* buf, unsigned* flags, int k, int stride, int iters, int work, int mode,
                                                            unsigned long long* cycles, int* err) {
    extern __shared__ float lds[];
    const int bid = blockIdx.x, tid = threadIdx.x;
    const int span = k * stride;
    const int group = (bid / span) * stride + (bid % stride), member = (bid % span) / stride;
    const int n_groups = gridDim.x / k;
    const int part = TILE_FLOATS / k;                        // floats this workgroup publishes per exchange
    float* gbuf = buf + (size_t)group * 2 * TILE_FLOATS;     // double-buffered tile of the group
    unsigned* gflag = flags + (size_t)group * 64;            // one flag word per member (separate 64-B apart would not matter at this k)
    float acc = (float)tid;
    __syncthreads();
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        float* tile = gbuf + (it & 1) * TILE_FLOATS;
        // "compute": a dependent chain standing in for the layer's K loop (0 = exchange cost alone)
        for (int w = 0; w < work; ++w) acc = acc * 1.0000001f + 0.5f;
        // payload that the readers can check: (iteration, publishing member, index) -> value
        for (int i = tid; i < part; i += THREADS) tile[member * part + i] = (float)((it & 1023) * 64 + member * 8 + (i & 7)) + acc * 0.f;
        if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // this wave's stores have left for L2
        __syncthreads();
        if (tid == 0) {
            if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(gflag + member, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < k && tid != member) {
            int spins = 0;
            while (__hip_atomic_load(gflag + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                if (++spins > 20000000) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (mode == 1 && tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float s = 0.f;
        for (int i = tid; i < TILE_FLOATS; i += THREADS) {                   // read the whole tile back (own part included) and CHECK it
            const float v = mode == 2 ? __hip_atomic_load(tile + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tile[i];
            const int m = i / part, j = i - m * part;
            if (v != (float)((it & 1023) * 64 + m * 8 + (j & 7))) atomicAdd(err + 1, 1);      // stale or torn data
            s += v;
        }
        acc += s * 1e-9f;
        if (*err) break;
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (tid == 0) { cycles[2 * bid] = t1 - t0; cycles[2 * bid + 1] = w1 - w0; }
    if (acc == 123.456f) lds[tid] = acc;
    (void)n_groups;
}

static void run(int wgs, int k, int stride, int iters, int work, int mode, double clock_hz, const char* tag) {
    float* buf; unsigned* flags; unsigned long long* cyc; int* err;
    const int groups = wgs / k;
    CK(hipMalloc(&buf, (size_t)groups * 2 * TILE_FLOATS * sizeof(float)));
    CK(hipMalloc(&flags, (size_t)groups * 64 * sizeof(unsigned)));
    CK(hipMalloc(&cyc, 2 * wgs * sizeof(unsigned long long)));
    CK(hipMalloc(&err, 2 * sizeof(int)));
    CK(hipMemset(flags, 0, (size_t)groups * 64 * sizeof(unsigned)));
    CK(hipMemset(err, 0, 2 * sizeof(int)));
    const size_t lds_bytes = 96 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(exchange_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(exchange_kernel, dim3(wgs), dim3(THREADS), lds_bytes, 0, buf, flags, k, stride, iters, work, mode, cyc, err);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(2 * wgs);
    int herr2[2] = {0, 0};
    CK(hipMemcpy(h.data(), cyc, 2 * wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    CK(hipMemcpy(herr2, err, 2 * sizeof(int), hipMemcpyDeviceToHost));
    const int herr = herr2[0];
    double mean = 0, mx = 0, wall = 0;
    for (int i = 0; i < wgs; ++i) { mean += (double)h[2 * i]; wall += (double)h[2 * i + 1]; if ((double)h[2 * i] > mx) mx = (double)h[2 * i]; }
    mean /= wgs; wall /= wgs;
    printf("%-22s mode %d  wgs %3d  k %d  stride %d  work %5d : %8.0f shader cycles per iteration (slowest workgroup %8.0f) = %.3f us%s\n", tag, mode,
           wgs, k, stride, work, mean / iters, mx / iters, wall / iters / clock_hz * 1e6, herr ? "  [FLAG LOST: spin bound hit]" : "");
    printf("%-22s                                                         stale / torn values read: %d of %lld\n", "", herr2[1],
           (long long)wgs * iters * TILE_FLOATS);
    CK(hipFree(buf)); CK(hipFree(flags)); CK(hipFree(cyc)); CK(hipFree(err));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double clock_hz = 100e6;                          // wall_clock64(): the constant 100 MHz counter; clock64(): shader cycles
    printf("%s  CUs %d\n", prop.name, prop.multiProcessorCount);
    const int iters = 2000;
    for (int mode : {0, 1, 2}) {
        run(64, 1, 8, iters, 0, mode, clock_hz, "no exchange (k = 1)");
        for (int k : {2, 4, 8}) {
            run(64, k, 8, iters, 0, mode, clock_hz, "same XCD");
            if (mode != 2) run(64, k, 1, iters, 0, mode, clock_hz, "across XCDs");
        }
        run(256, 2, 8, iters, 0, mode, clock_hz, "same XCD, chip full");
        run(256, 4, 8, iters, 0, mode, clock_hz, "same XCD, chip full");
    }
    // with a dependent chain between exchanges standing in for a split layer's K loop
    for (int mode : {1, 2}) {
        run(64, 2, 8, 400, 3000, mode, clock_hz, "same XCD + work");
        run(64, 4, 8, 400, 1500, mode, clock_hz, "same XCD + work");
    }
    run(64, 1, 8, 400, 3000, 2, clock_hz, "work alone (k = 1)");
    run(64, 1, 8, 400, 1500, 2, clock_hz, "work alone (k = 1)");
    return 0;
}
