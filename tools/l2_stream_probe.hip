// Probe: what does the L2 -> CU path deliver when EVERY CU re-reads the same weight stream?
//
// The v2 program kernel at one trajectory per workgroup streams the whole parameter set (15.8 MB of 1-KiB records for the config-2 net)
// through every CU once per solver step: DESIGN.md prices that against 64 B/clk/CU (34.5 TB/s over 256 CUs).  This probe measures the
// rate that access pattern reaches without any arithmetic, so the K loop of the program kernel can be priced against what the memory
// path delivers rather than against the datasheet figure.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/l2_stream_probe tools/l2_stream_probe.hip
//   tools/_bin/l2_stream_probe            # prints one line per variant: bytes per clock per CU, TB/s over the chip
//
// Variants: records in flight per wave (1..8), waves per workgroup (4 / 8 / 16), buffer size (2 MiB: stays in one XCD's L2; 15.8 MB: misses
// to the Infinity Cache once per XCD), and a per-workgroup rotation of the record order (all CUs of an XCD otherwise ask the same L2
// channel for the same line at the same time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const float4* __restrict__ buf, int n_records, int passes, int rotate, float* sink) {
    extern __shared__ float lds[];                       // sized by the host so that ONE workgroup fits per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int per_wave = n_records / nw;                 // records this wave reads per pass (n_records is a multiple of nw * DEPTH)
    const int start = rotate ? (int)(((long long)blockIdx.x * per_wave) / gridDim.x) / DEPTH * DEPTH : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < passes; ++p) {
        for (int i0 = 0; i0 < per_wave; i0 += DEPTH) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                int i = i0 + d + start;
                if (i >= per_wave) i -= per_wave;
                const int rec = i * nw + wave;            // records interleaved over the waves, as the program kernel deals them
                v[d] = buf[(size_t)rec * 64 + lane];
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[threadIdx.x] = acc.x + lds[threadIdx.x];   // keeps the loads alive
}

template <int DEPTH>
static void run(const float4* buf, size_t bytes, int waves, int passes, int rotate, int wgs, float* sink, double clock_hz, const char* tag) {
    int n_records = (int)(bytes / 1024);
    n_records -= n_records % (waves * DEPTH);
    const size_t lds_bytes = 96 * 1024;                  // > 80 KiB: a second workgroup does not fit beside it
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(wgs), dim3(waves * 64), lds_bytes, 0, buf, n_records, 2, rotate, sink);   // warm
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(wgs), dim3(waves * 64), lds_bytes, 0, buf, n_records, passes, rotate, sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double per_cu = (double)n_records * 1024.0 * passes;            // bytes each workgroup (= CU) pulled
    const double secs = best * 1e-3;
    printf("%-10s buffer %6.2f MB  wgs %3d  waves %2d  in flight/wave %d  rotate %d : %6.3f ms  %5.1f B/clk/CU  %5.2f TB/s chip\n", tag,
           bytes / 1e6, wgs, waves, DEPTH, rotate, best, per_cu / secs / clock_hz, per_cu * wgs / secs / 1e12);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double clock_hz = prop.clockRate * 1e3;
    printf("%s  CUs %d  clock %.0f MHz\n", prop.name, prop.multiProcessorCount, clock_hz / 1e6);
    const size_t big = 15800 * 1024, small = 2048 * 1024;
    float4* buf; float* sink;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&sink, 4096 * sizeof(float)));
    std::vector<float> h(big / 4, 1.0f);
    CK(hipMemcpy(buf, h.data(), big, hipMemcpyHostToDevice));
    const int cus = prop.multiProcessorCount;
    for (int rotate = 0; rotate < 2; ++rotate) {
        for (size_t bytes : {big, small}) {
            const int passes = bytes == big ? 20 : 160;
            const char* tag = bytes == big ? "stream" : "l2-fit";
            run<1>(buf, bytes, 8, passes, rotate, cus, sink, clock_hz, tag);
            run<2>(buf, bytes, 8, passes, rotate, cus, sink, clock_hz, tag);
            run<4>(buf, bytes, 8, passes, rotate, cus, sink, clock_hz, tag);
            run<8>(buf, bytes, 8, passes, rotate, cus, sink, clock_hz, tag);
            run<4>(buf, bytes, 4, passes, rotate, cus, sink, clock_hz, tag);
            run<4>(buf, bytes, 16, passes, rotate, cus, sink, clock_hz, tag);
            run<8>(buf, bytes, 16, passes, rotate, cus, sink, clock_hz, tag);
        }
    }
    // fewer workgroups: is the limit the CU's port or the shared L2?
    for (int wgs : {32, 64, 128}) run<4>(buf, big, 8, 20, 0, wgs, sink, clock_hz, "stream");
    return 0;
}
