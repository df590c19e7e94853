// cdx_optim.hip -- the optimiser side of DiffusionModel.update() as multi-tensor gfx950 kernels (first slice).
//
// What it replaces (reference cleandiffuser/diffusion/diffusionsde.py:114-141 + basic.py:66,83-86): per update() call
//   clip_grad_norm_(params, max_norm)  ->  ~3 ATen launches per parameter tensor + a stack/norm tail
//   torch.optim.AdamW.step()           ->  foreach kernels over ~100 small tensors (6-8 passes over the parameters)
//   optimizer.zero_grad()              ->  frees / re-allocates every .grad
//   ema_update()                       ->  2 ATen launches per parameter tensor
// Here: ONE table of (parameter, gradient, moment, EMA) pointers cut into 4096-float chunks, one workgroup per chunk,
//   mode SUMSQ  : per-chunk sum of g^2 -> fixed-order reduction -> total norm and clip coefficient on the device
//   mode ADAMW  : g*clip -> decoupled weight decay -> moments -> bias-corrected step -> optional EMA of the NEW parameter ->
//                 optional g <- 0, all in one pass: 5 streams read, 5 written, 40 B per parameter, HBM-bound
//   mode EMA    : ema <- rate*ema + (1-rate)*p alone (callers that update the EMA on their own schedule)
//   mode ZERO   : g <- 0
// Arithmetic follows torch.optim.AdamW's single-tensor path term by term (torch/optim/adamw.py: mul_(1 - lr*wd), lerp_, mul_/addcmul_,
// sqrt/bias_correction2_sqrt + eps, addcdiv_) so that parity with the PyTorch optimiser is rounding-level.
// Bound: HBM (8 TB/s).  No LDS tiling is needed: every element is touched once; the only cross-lane work is the norm reduction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "cdx.h"

extern void cdx_set_err(const char* msg);
#define cdx_set_error(...)                            \
    do {                                              \
        char buf_[256];                               \
        snprintf(buf_, sizeof(buf_), __VA_ARGS__);    \
        cdx_set_err(buf_);                            \
    } while (0)

namespace {

constexpr int OPT_THREADS = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one workgroup per chunk; the chunk never straddles tensors
template <int MODE>
__global__ __launch_bounds__(OPT_THREADS) void cdx_optim_kernel(cdx_optim_args a) {
    const int c = blockIdx.x;
    const int t = a.chunks[2 * c], ci = a.chunks[2 * c + 1];
    const long long n = a.numel[t];
    const long long base = (long long)ci * a.chunk_elems;
    const int len = (int)((n - base < a.chunk_elems) ? (n - base) : a.chunk_elems);
    constexpr bool STEP = MODE == CDX_OPT_ADAMW || MODE == CDX_OPT_ADAM;
    float* __restrict__ p = (STEP || MODE == CDX_OPT_EMA) ? a.p[t] + base : nullptr;
    float* __restrict__ g = (STEP || MODE == CDX_OPT_SUMSQ || MODE == CDX_OPT_ZERO) ? a.g[t] + base : nullptr;
    float* __restrict__ m = STEP ? a.m[t] + base : nullptr;
    float* __restrict__ v = STEP ? a.v[t] + base : nullptr;
    float* __restrict__ e = ((STEP && a.ema != nullptr) || MODE == CDX_OPT_EMA) ? a.ema[t] + base : nullptr;

    if (MODE == CDX_OPT_SUMSQ) {
        float s = 0.f;
        const bool vec = ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
        if (vec) {
            const float4* g4 = reinterpret_cast<const float4*>(g);
            for (int i = threadIdx.x; i < (len >> 2); i += OPT_THREADS) {
                float4 x = g4[i];
                s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            }
            for (int i = (len & ~3) + threadIdx.x; i < len; i += OPT_THREADS) s += g[i] * g[i];
        } else {
            for (int i = threadIdx.x; i < len; i += OPT_THREADS) s += g[i] * g[i];
        }
        __shared__ float red[OPT_THREADS / 64];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) a.partial[c] = (red[0] + red[1]) + (red[2] + red[3]);
        return;
    }
    if (MODE == CDX_OPT_ZERO) {
        for (int i = threadIdx.x; i < len; i += OPT_THREADS) g[i] = 0.f;
        return;
    }
    if (MODE == CDX_OPT_EMA) {
        const float r = a.ema_rate, q = 1.f - a.ema_rate;
        for (int i = threadIdx.x; i < len; i += OPT_THREADS) e[i] = e[i] * r + q * p[i];
        return;
    }
    // ADAMW (decoupled decay: p *= 1 - lr * wd) / ADAM (torch.optim.Adam's L2 form: g += wd * p, what the reference's classifiers
    // train with -- classifier/base.py:24)
    const float clip = (a.max_norm > 0.f) ? a.norm[1] : 1.f;
    const float decay = MODE == CDX_OPT_ADAM ? 1.f : 1.f - a.lr * a.weight_decay;
    const float l2 = MODE == CDX_OPT_ADAM ? a.weight_decay : 0.f;
    const float w1 = 1.f - a.beta1, w2 = 1.f - a.beta2;
    const float r = a.ema_rate, q = 1.f - a.ema_rate;
    const bool vec = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(e)) & 15) == 0);
    auto one = [&](float pp, float gg, float& mm, float& vv) -> float {
        gg *= clip;
        if (MODE == CDX_OPT_ADAM) gg = gg + l2 * pp;     // grad.add(param, alpha = weight_decay)
        else pp *= decay;
        mm = mm + w1 * (gg - mm);                       // lerp_(grad, 1 - beta1), weight < 0.5 branch
        vv = vv * a.beta2 + w2 * gg * gg;               // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
        return pp - a.step_size * (mm / denom);         // addcdiv_(exp_avg, denom, value = -step_size)
    };
    if (vec) {
        float4* p4 = reinterpret_cast<float4*>(p);
        float4* g4 = reinterpret_cast<float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        float4* e4 = reinterpret_cast<float4*>(e);
        for (int i = threadIdx.x; i < (len >> 2); i += OPT_THREADS) {
            float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
            pp.x = one(pp.x, gg.x, mm.x, vv.x);
            pp.y = one(pp.y, gg.y, mm.y, vv.y);
            pp.z = one(pp.z, gg.z, mm.z, vv.z);
            pp.w = one(pp.w, gg.w, mm.w, vv.w);
            p4[i] = pp, m4[i] = mm, v4[i] = vv;
            if (e != nullptr) {
                float4 ee = e4[i];
                ee.x = ee.x * r + q * pp.x, ee.y = ee.y * r + q * pp.y, ee.z = ee.z * r + q * pp.z, ee.w = ee.w * r + q * pp.w;
                e4[i] = ee;
            }
            if (a.zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int i = (vec ? (len & ~3) : 0) + threadIdx.x; i < len; i += OPT_THREADS) {
        float mm = m[i], vv = v[i];
        const float pn = one(p[i], g[i], mm, vv);
        p[i] = pn, m[i] = mm, v[i] = vv;
        if (e != nullptr) e[i] = e[i] * r + q * pn;
        if (a.zero_grad) g[i] = 0.f;
    }
}

// fixed-order reduction of the per-chunk partial sums: one workgroup, each thread a strided slice in ascending order, then a
// tree over the 256 thread sums -- the same bits for the same chunk table whatever the launch timing
__global__ __launch_bounds__(OPT_THREADS) void cdx_optim_norm_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                                     float* __restrict__ norm) {
    __shared__ float red[OPT_THREADS];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += OPT_THREADS) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = OPT_THREADS / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = sqrtf(red[0]);
        norm[0] = total;
        const float coef = max_norm / (total + 1e-6f);       // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max = 1)
        norm[1] = (max_norm > 0.f && coef < 1.f) ? coef : 1.f;
    }
}

}  // namespace

extern "C" int cdx_optim_f32(const cdx_optim_args* a, void* hip_stream) {
    if (a == nullptr) {
        cdx_set_error("cdx_optim_f32: null args");
        return CDX_EINVAL;
    }
    if (a->n_chunks == 0 || a->n_tensors == 0) return CDX_OK;
    if (a->n_chunks < 0 || a->n_tensors < 0 || a->chunk_elems <= 0 || (a->chunk_elems & 3) || a->chunks == nullptr || a->numel == nullptr) {
        cdx_set_error("cdx_optim_f32: bad chunk table (n_chunks %d, n_tensors %d, chunk_elems %d must be a positive multiple of 4)",
                      a->n_chunks, a->n_tensors, a->chunk_elems);
        return CDX_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const dim3 grid(a->n_chunks), block(OPT_THREADS);
    switch (a->mode) {
    case CDX_OPT_ADAMW:
    case CDX_OPT_ADAM:
        if (!a->p || !a->g || !a->m || !a->v || (a->max_norm > 0.f && !a->norm)) {
            cdx_set_error("cdx_optim_f32: AdamW needs the p / g / m / v tables (and norm when clipping): null pointer");
            return CDX_EINVAL;
        }
        if (!(a->bc2_sqrt > 0.f) || !(a->beta1 >= 0.f && a->beta1 < 1.f) || !(a->beta2 >= 0.f && a->beta2 < 1.f)) {
            cdx_set_error("cdx_optim_f32: betas must lie in [0, 1) and bc2_sqrt must be positive");
            return CDX_EINVAL;
        }
        if (a->mode == CDX_OPT_ADAM) hipLaunchKernelGGL(cdx_optim_kernel<CDX_OPT_ADAM>, grid, block, 0, st, *a);
        else hipLaunchKernelGGL(cdx_optim_kernel<CDX_OPT_ADAMW>, grid, block, 0, st, *a);
        break;
    case CDX_OPT_EMA:
        if (!a->p || !a->ema) {
            cdx_set_error("cdx_optim_f32: EMA needs the p and ema tables: null pointer");
            return CDX_EINVAL;
        }
        hipLaunchKernelGGL(cdx_optim_kernel<CDX_OPT_EMA>, grid, block, 0, st, *a);
        break;
    case CDX_OPT_SUMSQ:
        if (!a->g || !a->partial || !a->norm) {
            cdx_set_error("cdx_optim_f32: the norm pass needs the g table, partial[n_chunks] and norm[2]: null pointer");
            return CDX_EINVAL;
        }
        hipLaunchKernelGGL(cdx_optim_kernel<CDX_OPT_SUMSQ>, grid, block, 0, st, *a);
        hipLaunchKernelGGL(cdx_optim_norm_kernel, dim3(1), block, 0, st, a->partial, a->n_chunks, a->max_norm, a->norm);
        break;
    case CDX_OPT_ZERO:
        if (!a->g) {
            cdx_set_error("cdx_optim_f32: zeroing needs the g table: null pointer");
            return CDX_EINVAL;
        }
        hipLaunchKernelGGL(cdx_optim_kernel<CDX_OPT_ZERO>, grid, block, 0, st, *a);
        break;
    default:
        cdx_set_error("cdx_optim_f32: unknown mode %d", a->mode);
        return CDX_EINVAL;
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        cdx_set_error("cdx_optim_f32: %s", hipGetErrorString(err));
        return CDX_EHIP;
    }
    return CDX_OK;
}
