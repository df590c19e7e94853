// cdx_common.hip -- what every translation unit of libcdx.so shares: the per-thread error text, the ABI version and the MFMA
// lane-layout probe the tests run on silicon before they trust a record packing (include/cdx.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/cdx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static thread_local char g_err[256] = "";
void cdx_set_err(const char* msg) {          // shared with the other translation units of libcdx.so
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
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

// one wave per workgroup: which XCDs does the dispatcher reach from this process (cdx_device_query)
__global__ void cdx_xcc_probe_kernel(uint32_t* mask) {
    if (threadIdx.x == 0) atomicOr(mask, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u));     // hwreg(HW_REG_XCC_ID, 0, 4)
}

extern "C" {

int cdx_abi_version(void) { return CDX_ABI_VERSION; }

int cdx_device_query(int device, uint32_t* scratch_u32, void* hip_stream, cdx_device_props* out) {
    g_err[0] = 0;
    if (!scratch_u32 || !out) { cdx_set_err("cdx_device_query: null argument"); return CDX_EINVAL; }
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    memset(out, 0, sizeof(*out));
    out->cu_count = p.multiProcessorCount;
    out->lds_bytes_per_cu = (int32_t)p.maxSharedMemoryPerMultiProcessor;
    out->wavefront = p.warpSize;
    size_t i = 0;
    for (; i + 1 < sizeof(out->arch) && p.gcnArchName[i] && p.gcnArchName[i] != ':'; ++i) out->arch[i] = p.gcnArchName[i];
    out->arch[i] = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    e = hipMemsetAsync(scratch_u32, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cdx_xcc_probe_kernel, dim3(2048), dim3(64), 0, s, scratch_u32);
        e = hipGetLastError();
    }
    uint32_t mask = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&mask, scratch_u32, sizeof(mask), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    out->xcc_count = __builtin_popcount(mask);
    return CDX_OK;
}

const char* cdx_last_error(void) { return g_err; }

int cdx_probe_mfma_layout(float* out_device, void* hip_stream) {
    g_err[0] = 0;
    if (!out_device) { cdx_set_err("null output"); return CDX_EINVAL; }
    hipLaunchKernelGGL(cdx_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(hip_stream), out_device);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

}  // extern "C"
