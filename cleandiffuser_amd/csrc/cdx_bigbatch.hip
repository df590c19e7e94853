// cdx_bigbatch.hip -- whole sampling loops for the GEMM-shaped denoisers (DiT1d, residual MLPs) on gfx950.
//
// The one-workgroup-per-trajectory program kernel (cdx_unet2.hip) is the right shape while a trajectory's
// activations fit one CU's LDS and the batch is a few hundred.  DiT1d (64 tokens x 320..1280 features) and the
// wide IDQLMlp (hidden 1024..4096, 10^5..10^6 samples) are the opposite regime: M = batch x tokens is huge, every layer
// is a plain (M, K) x (K, N) GEMM, weights are re-used by every row.  Here the host sequences tiled-GEMM /
// LayerNorm / attention launches (csrc/cdx_gemm.hip) and a fused solver-step kernel; everything is enqueued on the
// caller's stream without synchronisation, so the loop of `sample()` (reference diffusion/diffusionsde.py:526-594,
// newedm.py:387-401) costs one C call.  The batch is processed in independent chunks sized so that a chunk's
// activations stay inside the 256 MiB Infinity Cache between producer and consumer launches.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>

#include "../../include/cdx.h"
#include "cdx_ops2.h"

extern void cdx_set_err(const char* msg);

// library-internal entries of csrc/cdx_gemm.hip (deferred split-K reduction, summed by the GroupNorm that consumes the conv)
int cdx_gemm_partials_f32(const cdx_gemm_args* g, void* hip_stream, int* slices);
bool cdx_groupnorm_slices_ok(int L, int C, int G);
int cdx_groupnorm_slices_f32(const cdx_gn_args* a, int slices, long long slice_stride, const float* xbias, void* hip_stream);

namespace {

// EDM / consistency records evaluate the network on c_in * x (reference newedm.py:142-148, edm.py:77-82): scaled copy of the state
__global__ void scale_rows_kernel(float* __restrict__ out, const float* __restrict__ x, float c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = c * x[i];
}

int scaled_input(hipStream_t st, const float*& x, float* scratch, float in_scale, size_t n) {
    if (in_scale == 1.0f) return CDX_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, st, scratch, x, in_scale, n);
    x = scratch;
    return hipGetLastError() == hipSuccess ? CDX_OK : CDX_EHIP;
}

// ------------------------------------------------------------------------------------------------
// Embedding rows of one chunk:  out[r] = temb[row(r)] + (conditional half ? cond[b0 + r] : 0)
// (reference dit.py:127-131: emb = map_noise(t); emb += condition | zeros)
// ------------------------------------------------------------------------------------------------
// All step records of a chunk at once: row (rec * rows + r).  The embedding path does not depend on the state x, so the
// whole map_emb / adaLN chain of every denoising step is evaluated BEFORE the loop as a few tall GEMMs.
__global__ void emb_rows_kernel(float* __restrict__ out, const float* __restrict__ temb, const float* __restrict__ cond,
                                int n_rec, int rows, int nb, int b0, int E, int per_sample, int n_cond_rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_rec * rows * E) return;
    const int e = (int)(i % E);
    const int rr = (int)(i / E);
    const int rec = rr / rows, r = rr - rec * rows;
    const int s = b0 + (r % nb);
    float v = temb[(size_t)(per_sample ? s : rec) * E + e];
    if (cond != nullptr && r < n_cond_rows) v += cond[(size_t)s * E + e];
    out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// Feature rows of the residual MLP:  [in_scale * x | tfeat | obs-or-zeros]   (reference idqlmlp.py:57-63)
// ------------------------------------------------------------------------------------------------
__global__ void mlp_features_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ temb,
                                    const float* __restrict__ cond, int rows, int nb, int b0, int D, int E, int O,
                                    int rec, int per_sample, int n_cond_rows, float in_scale) {
    const int F = D + E + O;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * F) return;
    const int r = (int)(i / F), c = (int)(i - (size_t)r * F);
    const int l = r % nb, s = b0 + l;
    float v;
    if (c < D) v = in_scale * x[(size_t)l * D + c];
    else if (c < D + E) v = temb[(size_t)(per_sample ? s : rec) * E + (c - D)];
    else v = (cond != nullptr && r < n_cond_rows) ? cond[(size_t)s * O + (c - D - E)] : 0.f;
    out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// One solver step on a chunk, state in HBM: guidance combine, clip, eps/x0 conversion, update, fix-mask blend.
// Same arithmetic, in the same order, as the in-LDS step of cdx_unet2.hip (kinds 0-4); kinds 5/6 are EDM.
// ------------------------------------------------------------------------------------------------
struct StepArgs {
    float* x;             // (nb, hd) chunk state, in/out
    const float* pred;    // (nb | 2 nb, hd)
    float* prev;          // (nb, hd) multistep memory / EDM slope
    float* xold;          // (nb, hd) EDM Heun: state before the predictor
    const float* prior;   // chunk-offset already applied
    const float* fix_mask;
    const float* noise;   // full tensor [n_noise][batch][hd]
    const float* x_min;
    const float* x_max;
    cdx_step st;
    int nb, hd, b0, batch, predict_noise, cfg_mode;
    float cfg_w;
    const float* grad;    // classifier gradient (nb, hd) or NULL
    float cg_scale;       // pred += cg_scale * grad before clipping (classifier guidance)
};

__global__ void solver_step_kernel(const StepArgs a) {
    const size_t n = (size_t)a.nb * a.hd;
    const cdx_step& st = a.st;
    const float al = st.alpha, sg = st.sigma;
    const float k0 = st.k[0], k1 = st.k[1], k2 = st.k[2], k3 = st.k[3], k4 = st.k[4];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % a.hd);
        const int l = (int)(i / a.hd);
        const float x = a.x[i];
        float p = a.pred[i];
        if (a.cfg_mode == 2) p = a.cfg_w * p + (1.0f - a.cfg_w) * a.pred[n + i];
        if (a.grad) p += a.cg_scale * a.grad[i];
        float xn;
        if (st.kind >= 5) {  // EDM: p is the raw network output F
            float d = k0 * x + k1 * p;
            if (a.x_min) d = fmaxf(d, a.x_min[e]);
            if (a.x_max) d = fminf(d, a.x_max[e]);
            if (st.kind == 7) {                          // consistency model: x <- f(x) [mask], then re-noise for the next level
                xn = d;
                if (a.fix_mask) {
                    const float m = a.fix_mask[e];
                    xn = xn * (1.0f - m) + a.prior[i] * m;
                }
                if (st.noise_idx >= 0) xn += k3 * a.noise[((size_t)st.noise_idx * a.batch + a.b0 + l) * a.hd + e];
                a.x[i] = xn;
                continue;
            }
            const float s = (x - d) / k2;
            if (st.kind == 5) {
                xn = x - s * k3;
                if (st.push) { a.prev[i] = s; a.xold[i] = x; }
            } else {
                xn = a.xold[i] - (a.prev[i] + s) / 2.0f * k3;
            }
        } else {
            if (a.predict_noise) {
                if (a.x_max) p = fmaxf(p, (x - al * a.x_max[e]) / sg);
                if (a.x_min) p = fminf(p, (x - al * a.x_min[e]) / sg);
            } else {
                if (a.x_min) p = fmaxf(p, a.x_min[e]);
                if (a.x_max) p = fminf(p, a.x_max[e]);
            }
            float eps, xth;
            if (a.predict_noise) { eps = p; xth = (x - sg * p) / al; }
            else { xth = p; eps = (x - al * p) / sg; }
            const size_t zi = ((size_t)(st.noise_idx < 0 ? 0 : st.noise_idx) * a.batch + a.b0 + l) * a.hd + e;
            if (st.kind >= 3) {
                const float m = a.fix_mask ? a.fix_mask[e] : 0.f;
                if (st.kind == 3) { p = p * (1.0f - m); xn = k0 * (x - k1 * p); }
                else { p = p * (1.0f - m) + x * m; xn = k0 * (k1 * x + k2 * p); }
                if (st.noise_idx >= 0) xn += k3 * a.noise[zi];
            } else if (st.kind == 0) {
                xn = k0 * (x - k1 * eps) + k2 * eps;
                if (st.noise_idx >= 0) xn += k3 * a.noise[zi];
            } else if (st.kind == 1) {
                xn = k0 * ((x - k1 * eps) / k2) + k3 * eps;
            } else {
                if (st.flags & CDX_STEP_MASK_PRED) {
                    const float m = a.fix_mask ? a.fix_mask[e] : 0.f;
                    eps = eps * (1.0f - m);
                    xth = xth * (1.0f - m) + x * m;
                }
                float v = (st.vsel & 1) ? xth : eps;
                if (st.vsel == 2) v = k3 * xth - k4 * a.prev[i];
                if (st.vsel == 3) v = k3 * eps - k4 * a.prev[i];
                xn = k0 * x - k1 * v;
                if (st.noise_idx >= 0) xn += k2 * a.noise[zi];
            }
            if (st.push) a.prev[i] = st.push == 2 ? eps : xth;
        }
        if (a.fix_mask) {
            const float m = a.fix_mask[e];
            xn = xn * (1.0f - m) + a.prior[i] * m;
        }
        a.x[i] = xn;
    }
}

// ------------------------------------------------------------------------------------------------
// host-side helpers
// ------------------------------------------------------------------------------------------------
struct Arena {
    float* base;
    long long used, cap;
    float* take(long long n) {
        n = (n + 63) & ~63LL;  // 256-byte aligned blocks: every buffer stays float4-loadable
        float* p = base ? base + used : nullptr;
        used += n;
        return p;
    }
};

#define CDX_TRY(expr)            \
    do {                         \
        const int rc_ = (expr);  \
        if (rc_ != CDX_OK) return rc_; \
    } while (0)

int hip_ok() {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cdx_set_err(hipGetErrorString(e)); return CDX_EHIP; }
    return CDX_OK;
}

int gemm(void* stream, const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N,
         int K, int act = CDX_ACT_NONE, const float* gate = nullptr, int ldg = 0, int rows_per_gate = 1,
         const float* residual = nullptr, int ldr = 0, const float* table = nullptr, int table_rows = 0) {
    cdx_gemm_args g;
    g.A = A; g.W = W; g.bias = bias; g.gate = gate; g.residual = residual; g.table = table; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldg = ldg; g.ldr = ldr;
    g.rows_per_gate = rows_per_gate; g.table_rows = table_rows; g.act = act;
    g.conv_taps = g.conv_cin = g.conv_lin = g.conv_lout = g.conv_stride = g.conv_pad = 0;
    g.partial = nullptr; g.partial_slices = 0;
    return cdx_gemm_f32(&g, stream);
}

int layernorm(void* stream, const float* x, float* y, int M, int C, float eps, const float* gamma, const float* beta,
              const float* scale, const float* shift, int ldmod, int rows_per_mod, int x_rows) {
    cdx_ln_args a;
    a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.scale = scale; a.shift = shift;
    a.M = M; a.C = C; a.ldx = C; a.ldy = C; a.ldmod = ldmod; a.rows_per_mod = rows_per_mod; a.eps = eps; a.x_rows = x_rows;
    return cdx_layernorm_f32(&a, stream);
}

int check_request(const cdx_sampling* s, const char* who, int max_kind) {
    if (!s || !s->x_in || !s->x_out || !s->temb) { cdx_set_err("null pointer in sampling request"); return CDX_EINVAL; }
    if (s->batch < 0 || s->hd <= 0 || s->emb_dim <= 0 || s->n_steps < 0 || s->chunk < 0) { cdx_set_err("bad size in sampling request"); return CDX_EINVAL; }
    if (s->n_steps > 0 && !s->steps) { cdx_set_err("steps == NULL with n_steps > 0"); return CDX_EINVAL; }
    if (s->fix_mask && !s->prior) { cdx_set_err("fix_mask given without prior"); return CDX_EINVAL; }
    if (s->cfg_mode < 0 || s->cfg_mode > 2) { cdx_set_err("cfg_mode must be 0, 1 or 2"); return CDX_EINVAL; }
    if (s->cfg_mode == 2 && (!s->cond || s->n_steps == 0)) { cdx_set_err("cfg_mode 2 needs cond and a sampling loop"); return CDX_EINVAL; }
    if (s->temb_per_sample && s->n_steps > 0) { cdx_set_err("per-sample timesteps are a forward-mode feature"); return CDX_EINVAL; }
    for (int i = 0; i < s->n_steps; ++i) {
        const cdx_step& st = s->steps[i];
        if (st.kind < 0 || st.kind > max_kind) { cdx_set_err("unsupported step kind for this executor"); return CDX_EINVAL; }
        if (st.noise_idx >= 0 && !s->noise) { cdx_set_err("step draws noise but noise == NULL"); return CDX_EINVAL; }
    }
    (void)who;
    return CDX_OK;
}

int chunk_of(const cdx_sampling* s) { return (s->chunk > 0 && s->chunk < s->batch) ? s->chunk : s->batch; }

int run_step(hipStream_t stream, const cdx_sampling* s, const cdx_step& st, float* x, const float* pred, float* prev,
             float* xold, int nb, int b0) {
    StepArgs a;
    a.x = x; a.pred = pred; a.prev = prev; a.xold = xold;
    a.prior = s->prior ? s->prior + (size_t)b0 * s->hd : nullptr;
    a.fix_mask = s->fix_mask; a.noise = s->noise; a.x_min = s->x_min; a.x_max = s->x_max;
    a.st = st; a.nb = nb; a.hd = s->hd; a.b0 = b0; a.batch = s->batch;
    a.predict_noise = s->predict_noise; a.cfg_mode = s->cfg_mode; a.cfg_w = s->cfg_w;
    a.grad = nullptr; a.cg_scale = 0.f;
    const size_t n = (size_t)nb * s->hd;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(solver_step_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hip_ok();
}

// ------------------------------------------------------------------------------------------------
// DiT1d
// ------------------------------------------------------------------------------------------------
struct DitBuffers {
    float *x, *prev, *xold, *emb0, *e1, *emb, *semb, *ada, *h0, *xm, *qkv, *att, *h2, *f, *h, *pred, *r0, *hc;
};

// DiT1Ref: [x_ref | x] state rows -> the reference half of the network output is the reference half of its input (dit.py:157,180)
__global__ void copy_ref_cols_kernel(float* __restrict__ out, const float* __restrict__ x, size_t rows, size_t x_rows, int in_dim,
                                     int width) {
    const size_t n = rows * (size_t)in_dim;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / in_dim, c = i - r * in_dim;
        out[r * width + c] = x[(r % x_rows) * width + c];
    }
}

long long dit_layout(const cdx_dit1d_weights* w, const cdx_sampling* s, float* base, DitBuffers* B) {
    const long long nb = chunk_of(s), bf = nb * (s->cfg_mode == 2 ? 2 : 1);
    const long long T = w->tokens, d = w->d_model, rows = bf * T;
    const long long n_rec = s->n_steps > 0 ? s->n_steps : 1, er = n_rec * bf;    // embedding rows: every record of the chunk
    Arena a{base, 0, 0};
    DitBuffers b;
    b.x = a.take(nb * s->hd);
    b.prev = a.take(nb * s->hd);
    b.xold = a.take(nb * s->hd);           // EDM Heun: state before the predictor
    b.emb0 = a.take(er * w->emb_dim);
    b.e1 = a.take(er * d);
    b.emb = a.take(er * d);
    b.semb = a.take(er * d);
    b.ada = a.take(er * (6 * d * w->depth + 2 * d));
    b.h0 = a.take(nb * T * d);
    b.xm = a.take(rows * d);
    b.qkv = a.take(rows * 3 * d);
    b.att = a.take(rows * d);
    b.h2 = a.take(rows * d);
    b.f = a.take(rows * 4 * d);
    b.h = a.take(rows * d);
    b.pred = a.take(rows * w->in_dim * (w->cross ? 2 : 1));
    b.r0 = b.hc = nullptr;
    if (w->cross) {
        b.r0 = a.take(nb * T * d);          // tokens of the reference half
        b.hc = a.take(rows * d);            // cross-attention output = the block's input
    }
    if (B) *B = b;
    return a.used;
}

int dit_check(const cdx_dit1d_weights* w, const cdx_sampling* s) {
    if (!w || !w->blocks || !w->x_proj_w || !w->pos || !w->map0_w || !w->map2_w || !w->fin_ada_w || !w->fin_w) { cdx_set_err("null pointer in DiT1d weights"); return CDX_EINVAL; }
    if (w->tokens <= 0 || w->tokens > CDX_ATTN_MAX_T || w->d_model <= 0 || w->d_model > 1024 || w->n_heads <= 0 ||
        w->d_model % w->n_heads != 0 || w->d_model / w->n_heads > 64 || w->depth < 0 || w->in_dim <= 0) {
        cdx_set_err("DiT1d executor: tokens <= 1024, d_model <= 1024, head_dim <= 64 required"); return CDX_EINVAL;
    }
    CDX_TRY(check_request(s, "cdx_dit1d_run", 7));
    if (w->cross) {
        for (int i = 0; i < w->depth; ++i)
            if (!w->cross[i].in_w || !w->cross[i].in_b || !w->cross[i].out_w || !w->cross[i].out_b) { cdx_set_err("null pointer in DiT1Ref cross-attention weights"); return CDX_EINVAL; }
    }
    if (s->hd != w->tokens * w->in_dim * (w->cross ? 2 : 1) || s->emb_dim != w->emb_dim || (s->cond && s->cond_dim != w->emb_dim)) {
        cdx_set_err("DiT1d request shape does not match the weights"); return CDX_EINVAL;
    }
    return CDX_OK;
}

// adaLN table of a chunk: for every step record, (bf, 6 d depth + 2 d) = [block 0: shift_a scale_a gate_a shift_m scale_m gate_m |
// block 1 ... | final: shift scale]   (reference dit.py:31, 47 evaluated once per record instead of once per block call)
int dit_prepare(const cdx_dit1d_weights* w, const cdx_sampling* s, hipStream_t st, const DitBuffers& B, int nb, int b0) {
    const int two = s->cfg_mode == 2 ? 2 : 1, bf = nb * two;
    const int d = w->d_model, E = w->emb_dim, n_rec = s->n_steps > 0 ? s->n_steps : 1, er = n_rec * bf;
    const int ntot = 6 * d * w->depth + 2 * d;
    const int n_cond_rows = (s->cond == nullptr || s->cfg_mode == 0) ? 0 : nb;   // conditional half comes first
    {
        const long long n = (long long)er * E;
        hipLaunchKernelGGL(emb_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B.emb0, s->temb, s->cond, n_rec,
                           bf, nb, b0, E, s->temb_per_sample, n_cond_rows);
        CDX_TRY(hip_ok());
    }
    CDX_TRY(gemm(st, B.emb0, E, w->map0_w, E, w->map0_b, B.e1, d, er, d, E, CDX_ACT_MISH));
    CDX_TRY(gemm(st, B.e1, d, w->map2_w, d, w->map2_b, B.emb, d, er, d, d, CDX_ACT_MISH));
    CDX_TRY(cdx_act_f32(B.emb, B.semb, (long long)er * d, CDX_ACT_SILU, st));
    for (int i = 0; i < w->depth; ++i)
        CDX_TRY(gemm(st, B.semb, d, w->blocks[i].ada_w, d, w->blocks[i].ada_b, B.ada + (size_t)i * 6 * d, ntot, er, 6 * d, d));
    CDX_TRY(gemm(st, B.semb, d, w->fin_ada_w, d, w->fin_ada_b, B.ada + (size_t)w->depth * 6 * d, ntot, er, 2 * d, d));
    return CDX_OK;
}

int dit_forward(const cdx_dit1d_weights* w, const cdx_sampling* s, hipStream_t st, const DitBuffers& B, const float* x,
                float* pred, int nb, int rec, float in_scale = 1.0f) {
    const int two = s->cfg_mode == 2 ? 2 : 1, bf = nb * two;
    const int T = w->tokens, d = w->d_model, rows = bf * T;
    const int xw = w->in_dim * (w->cross ? 2 : 1);                                 // width of a state row
    CDX_TRY(scaled_input(st, x, pred, in_scale, (size_t)nb * T * xw));            // pred is free until the final layer writes it
    const int ntot = 6 * d * w->depth + 2 * d;
    const float* ada_rec = B.ada + (size_t)rec * bf * ntot;
    // token stream: x_proj(x) + pos, computed once per trajectory (both CFG halves start from the same tokens)
    CDX_TRY(gemm(st, x + (w->cross ? w->in_dim : 0), xw, w->x_proj_w, w->in_dim, w->x_proj_b, B.h0, d, nb * T, d, w->in_dim, CDX_ACT_NONE,
                 nullptr, 0, 1, nullptr, 0, w->pos, T));
    if (w->cross)                                                                  // DiT1Ref: the reference half through the same projection
        CDX_TRY(gemm(st, x, xw, w->x_proj_w, w->in_dim, w->x_proj_b, B.r0, d, nb * T, d, w->in_dim, CDX_ACT_NONE, nullptr, 0, 1,
                     nullptr, 0, w->pos, T));
    const float* h = B.h0;
    int h_rows = nb * T;
    for (int i = 0; i < w->depth; ++i) {
        const cdx_dit1d_block& k = w->blocks[i];
        const float* ada = ada_rec + (size_t)i * 6 * d;
        if (w->cross) {
            // x <- MHA(q = x, k = v = x_ref tokens), no residual (dit.py:176).  q lands in columns [0, d) of the packed qkv rows,
            // k | v of the reference tokens in [d, 3d) -- the self-attention kernel then does the rest.  While both CFG halves still
            // share one token stream (first block) the attention runs once, on nb trajectories.
            const cdx_dit1ref_cross& c = w->cross[i];
            const int cb = h_rows / T;
            CDX_TRY(gemm(st, h, d, c.in_w, d, c.in_b, B.qkv, 3 * d, h_rows, d, d));
            for (int rep = 0; rep < cb / nb; ++rep)
                CDX_TRY(gemm(st, B.r0, d, c.in_w + (size_t)d * d, d, c.in_b + d, B.qkv + (size_t)rep * nb * T * 3 * d + d, 3 * d, nb * T,
                             2 * d, d));
            cdx_attn_args xa;
            xa.qkv = B.qkv; xa.out = B.att; xa.B = cb; xa.T = T; xa.n_heads = w->n_heads; xa.head_dim = d / w->n_heads;
            xa.scale = 1.0f / sqrtf((float)xa.head_dim); xa.mask = nullptr;
            CDX_TRY(cdx_attention_f32(&xa, st));
            CDX_TRY(gemm(st, B.att, d, c.out_w, d, c.out_b, B.hc, d, h_rows, d, d));
            h = B.hc;
        }
        // x <- modulate(LN(x), shift_a, scale_a)   (dit.py:33: the block continues from the modulated stream, Q4)
        CDX_TRY(layernorm(st, h, B.xm, rows, d, 1e-6f, nullptr, nullptr, ada + d, ada, ntot, T, h_rows == rows ? 0 : h_rows));
        CDX_TRY(gemm(st, B.xm, d, k.qkv_w, d, k.qkv_b, B.qkv, 3 * d, rows, 3 * d, d));
        cdx_attn_args at;
        at.qkv = B.qkv; at.out = B.att; at.B = bf; at.T = T; at.n_heads = w->n_heads; at.head_dim = d / w->n_heads;
        at.scale = 1.0f / sqrtf((float)at.head_dim); at.mask = nullptr;
        CDX_TRY(cdx_attention_f32(&at, st));
        CDX_TRY(gemm(st, B.att, d, k.proj_w, d, k.proj_b, B.h2, d, rows, d, d, CDX_ACT_NONE, ada + 2 * d, ntot, T, B.xm, d));
        CDX_TRY(layernorm(st, B.h2, B.xm, rows, d, 1e-6f, nullptr, nullptr, ada + 4 * d, ada + 3 * d, ntot, T, 0));
        CDX_TRY(gemm(st, B.xm, d, k.fc1_w, d, k.fc1_b, B.f, 4 * d, rows, 4 * d, d, CDX_ACT_GELU_TANH));
        CDX_TRY(gemm(st, B.f, 4 * d, k.fc2_w, 4 * d, k.fc2_b, B.h, d, rows, d, 4 * d, CDX_ACT_NONE, ada + 5 * d, ntot, T, B.h2, d));
        h = B.h;
        h_rows = rows;
    }
    const float* fin = ada_rec + (size_t)w->depth * 6 * d;
    CDX_TRY(layernorm(st, h, B.xm, rows, d, 1e-6f, nullptr, nullptr, fin + d, fin, ntot, T, h_rows == rows ? 0 : h_rows));
    CDX_TRY(gemm(st, B.xm, d, w->fin_w, d, w->fin_b, pred + (w->cross ? w->in_dim : 0), xw, rows, w->in_dim, d));
    if (w->cross) {       // (x may alias pred's first nb*T rows when the input was scaled: those rows then copy onto themselves)
        const size_t n = (size_t)rows * w->in_dim;
        hipLaunchKernelGGL(copy_ref_cols_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st,
                           pred, x, (size_t)rows, (size_t)nb * T, w->in_dim, xw);
        CDX_TRY(hip_ok());
    }
    return CDX_OK;
}

// ------------------------------------------------------------------------------------------------
// PearceTransformer (reference nn_diffusion/pearcetransformer.py:91-151) -- folded weights, see include/cdx.h
// ------------------------------------------------------------------------------------------------
struct PtfBuffers {
    float *x, *prev, *xold, *tin, *cin, *xe1, *xe, *xi, *f, *qkv, *att, *t1, *a1, *u, *f2, *pred;
};

long long ptf_layout(const cdx_pearcetf_weights* w, const cdx_sampling* s, float* base, PtfBuffers* B) {
    const long long nb = chunk_of(s), bf = nb * (s->cfg_mode == 2 ? 2 : 1);
    const long long S = 2 + w->To, te = w->te, td = (long long)w->te * w->n_heads, rows = bf * S;
    const long long n_rec = s->n_steps > 0 ? s->n_steps : 1;
    Arena a{base, 0, 0};
    PtfBuffers b;
    b.x = a.take(nb * s->hd);
    b.prev = a.take(nb * s->hd);
    b.xold = a.take(nb * s->hd);
    b.tin = a.take((s->temb_per_sample ? nb : n_rec) * te);
    b.cin = a.take(nb * w->To * te);
    b.xe1 = a.take(nb * w->emb_dim);
    b.xe = a.take(nb * w->emb_dim);
    b.xi = a.take(nb * te);
    b.f = a.take(rows * te);
    b.qkv = a.take(rows * 3 * td);
    b.att = a.take(rows * td);
    b.t1 = a.take(rows * te);
    b.a1 = a.take(rows * te);
    b.u = a.take(rows * 4 * te);
    b.f2 = a.take(rows * te);
    b.pred = a.take(bf * s->hd);
    if (B) *B = b;
    return a.used;
}

int ptf_check(const cdx_pearcetf_weights* w, const cdx_sampling* s) {
    if (!w || !w->blocks || !w->ae0_w || !w->ae0_b || !w->ae2_w || !w->ae2_b || !w->a2i_w || !w->a2i_b || !w->t2i_w || !w->t2i_b ||
        !w->c2i_w || !w->c2i_b || !w->cpos || !w->fin_w || !w->fin_b) { cdx_set_err("null pointer in PearceTransformer weights"); return CDX_EINVAL; }
    if (w->act_dim <= 0 || w->To <= 0 || 2 + w->To > 64 || w->emb_dim <= 0 || w->te <= 0 || w->te > 64 || w->n_heads <= 0 || w->n_blocks < 0) {
        cdx_set_err("PearceTransformer executor: 2 + To <= 64 tokens, trans_emb_dim (= head width) <= 64 required"); return CDX_EINVAL;
    }
    for (int i = 0; i < w->n_blocks; ++i) {
        const cdx_pearcetf_block& k = w->blocks[i];
        if (!k.qkv_w || !k.qkv_b || !k.o_w || !k.o_b || !k.r1 || !k.fc1_w || !k.fc1_b || !k.fc2_w || !k.fc2_b || !k.r2) {
            cdx_set_err("null pointer in a PearceTransformer block"); return CDX_EINVAL;
        }
    }
    CDX_TRY(check_request(s, "cdx_pearcetf_run", 7));
    if (s->hd != w->act_dim || s->emb_dim != w->emb_dim || s->cfg_mode == 0 || !s->cond || s->cond_dim != w->To * w->emb_dim) {
        cdx_set_err("PearceTransformer request: hd == act_dim, emb_dim, a (To, emb_dim) condition per sample (cfg_mode 1 or 2) required");
        return CDX_EINVAL;
    }
    return CDX_OK;
}

// y[r][c] = x[r][c] * scale[c]   (the residual paths after the folded BatchNorm1d)
__global__ void chan_scale_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ scale, size_t n, int C) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * scale[i % C];
}

// token rows of one forward: f[(r, 0)] = action token of sample r % nb, f[(r, 1)] = time token of the record (or of the sample),
// f[(r, 2 + j)] = condition token j of the sample (conditional half) or cond_to_input(0) + pos = c2i_b + cpos[j] (zero-condition half)
__global__ void ptf_tokens_kernel(float* __restrict__ f, const float* __restrict__ xi, const float* __restrict__ tin,
                                  const float* __restrict__ cin, const float* __restrict__ c2i_b, const float* __restrict__ cpos,
                                  int bf, int nb, int S, int te, int t_row, int per_sample, int n_cond_rows) {
    const size_t n = (size_t)bf * S * te;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % te);
        const int tok = (int)((i / te) % S);
        const int r = (int)(i / ((size_t)te * S));
        const int l = r % nb;
        float v;
        if (tok == 0) v = xi[(size_t)l * te + c];
        else if (tok == 1) v = tin[(size_t)(per_sample ? l : t_row) * te + c];
        else if (r < n_cond_rows) v = cin[((size_t)l * (S - 2) + (tok - 2)) * te + c];
        else v = c2i_b[c] + cpos[(size_t)(tok - 2) * te + c];
        f[i] = v;
    }
}

int ptf_prepare(const cdx_pearcetf_weights* w, const cdx_sampling* s, hipStream_t st, const PtfBuffers& B, int nb, int b0) {
    const int te = w->te, E = w->emb_dim;
    const int n_rec = s->n_steps > 0 ? s->n_steps : 1;
    // time tokens of every step record (or of every sample in forward mode), condition tokens of the chunk: independent of x
    if (s->temb_per_sample) CDX_TRY(gemm(st, s->temb + (size_t)b0 * E, E, w->t2i_w, E, w->t2i_b, B.tin, te, nb, te, E));
    else CDX_TRY(gemm(st, s->temb, E, w->t2i_w, E, w->t2i_b, B.tin, te, n_rec, te, E));
    CDX_TRY(gemm(st, s->cond + (size_t)b0 * s->cond_dim, E, w->c2i_w, E, w->c2i_b, B.cin, te, nb * w->To, te, E, CDX_ACT_NONE, nullptr, 0, 1,
                 nullptr, 0, w->cpos, w->To));
    return CDX_OK;
}

int ptf_forward(const cdx_pearcetf_weights* w, const cdx_sampling* s, hipStream_t st, const PtfBuffers& B, const float* x, float* pred,
                int nb, int rec, float in_scale = 1.0f) {
    const int two = s->cfg_mode == 2 ? 2 : 1, bf = nb * two;
    const int S = 2 + w->To, te = w->te, td = w->te * w->n_heads, E = w->emb_dim, rows = bf * S;
    CDX_TRY(scaled_input(st, x, pred, in_scale, (size_t)nb * w->act_dim));
    CDX_TRY(gemm(st, x, w->act_dim, w->ae0_w, w->act_dim, w->ae0_b, B.xe1, E, nb, E, w->act_dim, CDX_ACT_LEAKY));
    CDX_TRY(gemm(st, B.xe1, E, w->ae2_w, E, w->ae2_b, B.xe, E, nb, E, E));
    CDX_TRY(gemm(st, B.xe, E, w->a2i_w, E, w->a2i_b, B.xi, te, nb, te, E));
    {
        const size_t n = (size_t)rows * te;
        hipLaunchKernelGGL(ptf_tokens_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st, B.f, B.xi,
                           B.tin, B.cin, w->c2i_b, w->cpos, bf, nb, S, te, rec, s->temb_per_sample, nb /* cfg_mode >= 1: first nb rows */);
        CDX_TRY(hip_ok());
    }
    float* f = B.f;
    float* f_next = B.f2;
    const size_t n = (size_t)rows * te;
    const unsigned eb = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    for (int i = 0; i < w->n_blocks; ++i) {
        const cdx_pearcetf_block& k = w->blocks[i];
        CDX_TRY(gemm(st, f, te, k.qkv_w, te, k.qkv_b, B.qkv, 3 * td, rows, 3 * td, te));
        cdx_attn_args at;
        at.qkv = B.qkv; at.out = B.att; at.B = bf; at.T = S; at.n_heads = w->n_heads; at.head_dim = te;
        at.scale = 1.0f / sqrtf((float)te); at.mask = nullptr;
        CDX_TRY(cdx_attention_f32(&at, st));
        hipLaunchKernelGGL(chan_scale_kernel, dim3(eb), dim3(256), 0, st, B.t1, f, k.r1, n, te);
        CDX_TRY(hip_ok());
        CDX_TRY(gemm(st, B.att, td, k.o_w, td, k.o_b, B.a1, te, rows, te, td, CDX_ACT_NONE, nullptr, 0, 1, B.t1, te));
        CDX_TRY(gemm(st, B.a1, te, k.fc1_w, te, k.fc1_b, B.u, 4 * te, rows, 4 * te, te, CDX_ACT_GELU_ERF));
        hipLaunchKernelGGL(chan_scale_kernel, dim3(eb), dim3(256), 0, st, B.t1, B.a1, k.r2, n, te);
        CDX_TRY(hip_ok());
        CDX_TRY(gemm(st, B.u, 4 * te, k.fc2_w, 4 * te, k.fc2_b, f_next, te, rows, te, 4 * te, CDX_ACT_NONE, nullptr, 0, 1, B.t1, te));
        float* t = f; f = f_next; f_next = t;
    }
    CDX_TRY(gemm(st, f, S * te, w->fin_w, S * te, w->fin_b, pred, w->act_dim, bf, w->act_dim, S * te));
    return CDX_OK;
}

// ------------------------------------------------------------------------------------------------
// residual MLP
// ------------------------------------------------------------------------------------------------
struct MlpBuffers {
    float *x, *prev, *xold, *feat, *h, *y, *u, *pred;
};

long long mlp_layout(const cdx_resmlp_weights* w, const cdx_sampling* s, float* base, MlpBuffers* B) {
    const long long nb = chunk_of(s), bf = nb * (s->cfg_mode == 2 ? 2 : 1);
    const long long F = w->x_dim + w->emb_dim + w->obs_dim, H = w->hidden;
    Arena a{base, 0, 0};
    MlpBuffers b;
    b.x = a.take(nb * s->hd);
    b.prev = a.take(nb * s->hd);
    b.xold = a.take(nb * s->hd);
    b.feat = a.take(bf * F);
    b.h = a.take(bf * H);
    b.y = a.take(bf * H);
    b.u = a.take(bf * 4 * H);
    b.pred = a.take(bf * w->x_dim);
    if (B) *B = b;
    return a.used;
}

int mlp_check(const cdx_resmlp_weights* w, const cdx_sampling* s) {
    if (!w || !w->in_w || !w->out_w || (w->n_blocks > 0 && !w->blocks)) { cdx_set_err("null pointer in residual-MLP weights"); return CDX_EINVAL; }
    if (w->x_dim <= 0 || w->emb_dim <= 0 || w->obs_dim < 0 || w->hidden <= 0 || w->hidden > 4096 || w->n_blocks < 0) {
        cdx_set_err("residual-MLP executor: hidden <= 4096 required (LayerNorm row in registers)"); return CDX_EINVAL;
    }
    CDX_TRY(check_request(s, "cdx_resmlp_run", 7));
    if (s->hd != w->x_dim || s->emb_dim != w->emb_dim || (s->cond && s->cond_dim != w->obs_dim)) {
        cdx_set_err("residual-MLP request shape does not match the weights"); return CDX_EINVAL;
    }
    return CDX_OK;
}

int mlp_forward(const cdx_resmlp_weights* w, const cdx_sampling* s, hipStream_t st, const MlpBuffers& B, const float* x,
                float* pred, int nb, int b0, int rec, float in_scale) {
    const int two = s->cfg_mode == 2 ? 2 : 1, bf = nb * two;
    const int D = w->x_dim, E = w->emb_dim, O = w->obs_dim, F = D + E + O, H = w->hidden;
    const int n_cond_rows = (s->cond == nullptr || s->cfg_mode == 0) ? 0 : nb;
    {
        const long long n = (long long)bf * F;
        hipLaunchKernelGGL(mlp_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B.feat, x, s->temb,
                           s->cond, bf, nb, b0, D, E, O, rec, s->temb_per_sample, n_cond_rows, in_scale);
        CDX_TRY(hip_ok());
    }
    CDX_TRY(gemm(st, B.feat, F, w->in_w, F, w->in_b, B.h, H, bf, H, F));
    for (int i = 0; i < w->n_blocks; ++i) {
        const cdx_resmlp_block& k = w->blocks[i];
        CDX_TRY(layernorm(st, B.h, B.y, bf, H, 1e-5f, k.ln_g, k.ln_b, nullptr, nullptr, 0, 1, 0));
        CDX_TRY(gemm(st, B.y, H, k.fc1_w, H, k.fc1_b, B.u, 4 * H, bf, 4 * H, H, CDX_ACT_MISH));
        CDX_TRY(gemm(st, B.u, 4 * H, k.fc2_w, 4 * H, k.fc2_b, B.h, H, bf, H, 4 * H, CDX_ACT_NONE, nullptr, 0, 1, B.h, H));
    }
    const float* head_in = B.h;
    if (w->head_mish) {
        CDX_TRY(cdx_act_f32(B.h, B.y, (long long)bf * H, CDX_ACT_MISH, st));
        head_in = B.y;
    }
    CDX_TRY(gemm(st, head_in, H, w->out_w, H, w->out_b, pred, D, bf, D, H));
    return CDX_OK;
}

// ------------------------------------------------------------------------------------------------
// ChiTransformer
// ------------------------------------------------------------------------------------------------
// observation rows of a chunk: conditional samples copy their (To, obs_dim) block, unconditional ones are zeros
__global__ void obs_rows_kernel(float* __restrict__ out, const float* __restrict__ cond, int rows, int nb, int b0, int width,
                                int n_cond_rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * width) return;
    const int r = (int)(i / width), c = (int)(i - (size_t)r * width);
    out[i] = (cond != nullptr && r < n_cond_rows) ? cond[(size_t)(b0 + r % nb) * width + c] : 0.f;
}

// timestep tokens: temb[row] + cond_pos_emb[0]
__global__ void time_rows_kernel(float* __restrict__ out, const float* __restrict__ temb, const float* __restrict__ pos0,
                                 int rows, int nb, int b0, int d, int per_sample) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * d) return;
    const int r = (int)(i / d), c = (int)(i - (size_t)r * d);
    out[i] = temb[(size_t)(per_sample ? b0 + r % nb : r) * d + c] + pos0[c];
}

struct TfBuffers {
    float *x, *prev, *xold, *obs, *tin, *tenc, *tmem, *oin, *oenc, *omem, *tkv, *okv, *h, *y, *qkv, *att, *f, *pred;
    float *mem, *my, *mqkv, *matt, *mf;      // transformer condition encoder: (samples x (1 + To)) token rows
};

// memory tokens of a transformer condition encoder: row (r, 0) = the timestep token, rows (r, 1..To) = the observation tokens
__global__ void mem_rows_kernel(float* __restrict__ mem, const float* __restrict__ tin, const float* __restrict__ oin, int bf, int S,
                                int d, int trow_index, int per_sample) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)bf * S * d) return;
    const int c = (int)(i % d), j = (int)((i / d) % S), r = (int)(i / ((size_t)d * S));
    mem[i] = j == 0 ? tin[(size_t)(per_sample ? r : trow_index) * d + c] : oin[((size_t)r * (S - 1) + (j - 1)) * d + c];
}
// ... and back into the two blocks the cross-attention kernel reads (token 0 per sample | observation tokens)
__global__ void mem_split_kernel(const float* __restrict__ mem, float* __restrict__ tmem, float* __restrict__ omem, int bf, int S, int d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)bf * S * d) return;
    const int c = (int)(i % d), j = (int)((i / d) % S), r = (int)(i / ((size_t)d * S));
    if (j == 0) tmem[(size_t)r * d + c] = mem[i];
    else omem[((size_t)r * (S - 1) + (j - 1)) * d + c] = mem[i];
}

long long tf_layout(const cdx_chitf_weights* w, const cdx_sampling* s, float* base, TfBuffers* B) {
    const long long nb = chunk_of(s), bf = nb * (s->cfg_mode == 2 ? 2 : 1);
    const long long d = w->d_model, rows = bf * w->Ta, orow = bf * w->To;
    const long long trow = s->temb_per_sample ? bf : (s->n_steps > 0 ? s->n_steps : 1);
    const long long mtrow = w->n_enc_layers > 0 ? bf : trow;       // rows of the timestep token's memory / K / V
    const long long mrow = w->n_enc_layers > 0 ? bf * (1 + w->To) : 0;
    Arena a{base, 0, 0};
    TfBuffers b;
    b.x = a.take(nb * s->hd);
    b.prev = a.take(nb * s->hd);
    b.xold = a.take(nb * s->hd);
    b.obs = a.take(orow * w->obs_dim);
    b.tin = a.take(trow * d);
    b.tenc = a.take(trow * 4 * d);
    b.tmem = a.take(mtrow * d);
    b.oin = a.take(orow * d);
    b.oenc = a.take(orow * 4 * d);
    b.omem = a.take(orow * d);
    b.tkv = a.take((long long)w->n_layers * mtrow * 2 * d);
    b.okv = a.take((long long)w->n_layers * orow * 2 * d);
    b.h = a.take(rows * d);
    b.y = a.take(rows * d);
    b.qkv = a.take(rows * 3 * d);
    b.att = a.take(rows * d);
    b.f = a.take(rows * 4 * d);
    b.pred = a.take(rows * w->act_dim);
    b.mem = a.take(mrow * d);
    b.my = a.take(mrow * d);
    b.mqkv = a.take(mrow * 3 * d);
    b.matt = a.take(mrow * d);
    b.mf = a.take(mrow * 4 * d);
    if (B) *B = b;
    return a.used;
}

int tf_check(const cdx_chitf_weights* w, const cdx_sampling* s) {
    if (!w || !w->layers || !w->act_emb_w || !w->pos_emb || !w->obs_emb_w || !w->cond_pos_emb ||
        !w->lnf_g || !w->head_w || !w->self_mask || !w->memory_mask) { cdx_set_err("null pointer in ChiTransformer weights"); return CDX_EINVAL; }
    if (w->n_enc_layers < 0 || (w->n_enc_layers > 0 && !w->enc_layers) || (w->n_enc_layers == 0 && (!w->enc0_w || !w->enc2_w))) {
        cdx_set_err("ChiTransformer condition encoder: MLP weights (enc0 / enc2) or enc_layers required"); return CDX_EINVAL;
    }
    if (w->Ta <= 0 || w->Ta > 64 || w->To < 0 || 1 + w->To > 16 || w->d_model <= 0 || w->d_model > 1024 || w->n_heads <= 0 ||
        w->d_model % w->n_heads != 0 || w->d_model / w->n_heads > 64 || w->n_layers < 0) {
        cdx_set_err("ChiTransformer executor: Ta <= 64, 1 + To <= 16, d_model <= 1024, head_dim <= 64 required"); return CDX_EINVAL;
    }
    CDX_TRY(check_request(s, "cdx_chitf_run", 7));
    if (s->hd != w->Ta * w->act_dim || s->emb_dim != w->d_model || (s->cond && s->cond_dim != w->To * w->obs_dim)) {
        cdx_set_err("ChiTransformer request shape does not match the weights"); return CDX_EINVAL;
    }
    return CDX_OK;
}

// memory tokens and their per-layer K/V projections (independent of the state x)
int tf_prepare(const cdx_chitf_weights* w, const cdx_sampling* s, hipStream_t st, const TfBuffers& B, int nb, int b0) {
    const int two = s->cfg_mode == 2 ? 2 : 1, bf = nb * two, d = w->d_model, To = w->To;
    const int orow = bf * To, trow = s->temb_per_sample ? bf : (s->n_steps > 0 ? s->n_steps : 1);
    const int n_cond_rows = (s->cond == nullptr || s->cfg_mode == 0) ? 0 : nb * To;
    {
        const long long n = (long long)trow * d;
        hipLaunchKernelGGL(time_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B.tin, s->temb, w->cond_pos_emb,
                           trow, nb, b0, d, s->temb_per_sample);
        CDX_TRY(hip_ok());
    }
    const bool tfenc = w->n_enc_layers > 0;       // transformer encoder: the memory is built per step in tf_forward
    if (!tfenc) {
        CDX_TRY(gemm(st, B.tin, d, w->enc0_w, d, w->enc0_b, B.tenc, 4 * d, trow, 4 * d, d, CDX_ACT_MISH));
        CDX_TRY(gemm(st, B.tenc, 4 * d, w->enc2_w, 4 * d, w->enc2_b, B.tmem, d, trow, d, 4 * d));
    }
    if (To > 0) {
        const long long n = (long long)orow * w->obs_dim;
        // cond is (batch, To * obs_dim): row (b, s) of the observation table is a contiguous slice of it
        hipLaunchKernelGGL(obs_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B.obs, s->cond, bf, nb, b0,
                           To * w->obs_dim, n_cond_rows / (To > 0 ? To : 1));
        CDX_TRY(hip_ok());
        CDX_TRY(gemm(st, B.obs, w->obs_dim, w->obs_emb_w, w->obs_dim, w->obs_emb_b, B.oin, d, orow, d, w->obs_dim, CDX_ACT_NONE,
                     nullptr, 0, 1, nullptr, 0, w->cond_pos_emb + d, To));
        if (!tfenc) {
            CDX_TRY(gemm(st, B.oin, d, w->enc0_w, d, w->enc0_b, B.oenc, 4 * d, orow, 4 * d, d, CDX_ACT_MISH));
            CDX_TRY(gemm(st, B.oenc, 4 * d, w->enc2_w, 4 * d, w->enc2_b, B.omem, d, orow, d, 4 * d));
        }
    }
    if (tfenc) return CDX_OK;
    for (int l = 0; l < w->n_layers; ++l) {
        const cdx_chitf_layer& k = w->layers[l];
        CDX_TRY(gemm(st, B.tmem, d, k.ca_in_w + (size_t)d * d, d, k.ca_in_b + d, B.tkv + (size_t)l * trow * 2 * d, 2 * d, trow, 2 * d, d));
        if (To > 0)
            CDX_TRY(gemm(st, B.omem, d, k.ca_in_w + (size_t)d * d, d, k.ca_in_b + d, B.okv + (size_t)l * orow * 2 * d, 2 * d, orow, 2 * d, d));
    }
    return CDX_OK;
}

int tf_forward(const cdx_chitf_weights* w, const cdx_sampling* s, hipStream_t st, const TfBuffers& B, const float* x, float* pred,
               int nb, int rec, float in_scale = 1.0f) {
    const int two = s->cfg_mode == 2 ? 2 : 1, bf = nb * two;
    const int T = w->Ta, d = w->d_model, rows = bf * T, orow = bf * w->To;
    CDX_TRY(scaled_input(st, x, pred, in_scale, (size_t)nb * T * w->act_dim));    // pred is free until the head writes it
    int trow = s->temb_per_sample ? bf : (s->n_steps > 0 ? s->n_steps : 1);
    if (w->n_enc_layers > 0) {
        // nn.TransformerEncoder over [timestep token | observation tokens] of every sample (reference chitransformer.py:146-149),
        // then the decoder layers' K / V projections of that memory -- per step, because the timestep token takes part
        const int S = 1 + w->To, mrow = bf * S;
        const long long n = (long long)mrow * d;
        hipLaunchKernelGGL(mem_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B.mem, B.tin, B.oin, bf, S, d, rec,
                           s->temb_per_sample);
        CDX_TRY(hip_ok());
        for (int l = 0; l < w->n_enc_layers; ++l) {
            const cdx_chitf_enc_layer& k = w->enc_layers[l];
            CDX_TRY(layernorm(st, B.mem, B.my, mrow, d, 1e-5f, k.ln1_g, k.ln1_b, nullptr, nullptr, 0, 1, 0));
            CDX_TRY(gemm(st, B.my, d, k.sa_in_w, d, k.sa_in_b, B.mqkv, 3 * d, mrow, 3 * d, d));
            cdx_attn_args at;
            at.qkv = B.mqkv; at.out = B.matt; at.B = bf; at.T = S; at.n_heads = w->n_heads; at.head_dim = d / w->n_heads;
            at.scale = 1.0f / sqrtf((float)at.head_dim); at.mask = nullptr;
            CDX_TRY(cdx_attention_f32(&at, st));
            CDX_TRY(gemm(st, B.matt, d, k.sa_out_w, d, k.sa_out_b, B.mem, d, mrow, d, d, CDX_ACT_NONE, nullptr, 0, 1, B.mem, d));
            CDX_TRY(layernorm(st, B.mem, B.my, mrow, d, 1e-5f, k.ln2_g, k.ln2_b, nullptr, nullptr, 0, 1, 0));
            CDX_TRY(gemm(st, B.my, d, k.ff1_w, d, k.ff1_b, B.mf, 4 * d, mrow, 4 * d, d, CDX_ACT_GELU_ERF));
            CDX_TRY(gemm(st, B.mf, 4 * d, k.ff2_w, 4 * d, k.ff2_b, B.mem, d, mrow, d, 4 * d, CDX_ACT_NONE, nullptr, 0, 1, B.mem, d));
        }
        hipLaunchKernelGGL(mem_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B.mem, B.tmem, B.omem, bf, S, d);
        CDX_TRY(hip_ok());
        trow = bf;
        for (int l = 0; l < w->n_layers; ++l) {
            const cdx_chitf_layer& k = w->layers[l];
            CDX_TRY(gemm(st, B.tmem, d, k.ca_in_w + (size_t)d * d, d, k.ca_in_b + d, B.tkv + (size_t)l * trow * 2 * d, 2 * d, trow, 2 * d, d));
            if (w->To > 0)
                CDX_TRY(gemm(st, B.omem, d, k.ca_in_w + (size_t)d * d, d, k.ca_in_b + d, B.okv + (size_t)l * orow * 2 * d, 2 * d, orow, 2 * d, d));
        }
    }
    for (int half = 0; half < two; ++half)               // both CFG halves start from the same action tokens
        CDX_TRY(gemm(st, x, w->act_dim, w->act_emb_w, w->act_dim, w->act_emb_b, B.h + (size_t)half * nb * T * d, d, nb * T, d,
                     w->act_dim, CDX_ACT_NONE, nullptr, 0, 1, nullptr, 0, w->pos_emb, T));
    for (int l = 0; l < w->n_layers; ++l) {
        const cdx_chitf_layer& k = w->layers[l];
        CDX_TRY(layernorm(st, B.h, B.y, rows, d, 1e-5f, k.ln1_g, k.ln1_b, nullptr, nullptr, 0, 1, 0));
        CDX_TRY(gemm(st, B.y, d, k.sa_in_w, d, k.sa_in_b, B.qkv, 3 * d, rows, 3 * d, d));
        cdx_attn_args at;
        at.qkv = B.qkv; at.out = B.att; at.B = bf; at.T = T; at.n_heads = w->n_heads; at.head_dim = d / w->n_heads;
        at.scale = 1.0f / sqrtf((float)at.head_dim); at.mask = w->self_mask;
        CDX_TRY(cdx_attention_f32(&at, st));
        CDX_TRY(gemm(st, B.att, d, k.sa_out_w, d, k.sa_out_b, B.h, d, rows, d, d, CDX_ACT_NONE, nullptr, 0, 1, B.h, d));
        CDX_TRY(layernorm(st, B.h, B.y, rows, d, 1e-5f, k.ln2_g, k.ln2_b, nullptr, nullptr, 0, 1, 0));
        CDX_TRY(gemm(st, B.y, d, k.ca_in_w, d, k.ca_in_b, B.qkv, d, rows, d, d));       // q projection only
        cdx_xattn_args xa;
        xa.q = B.qkv; xa.kv_shared = B.tkv + (size_t)l * trow * 2 * d; xa.kv_rows = B.okv + (size_t)l * orow * 2 * d;
        xa.mask = w->memory_mask; xa.out = B.att; xa.B = bf; xa.T = T; xa.n_obs = w->To; xa.n_heads = w->n_heads;
        xa.head_dim = d / w->n_heads; xa.shared_row = rec; xa.shared_per_sample = (s->temb_per_sample || w->n_enc_layers > 0) ? 1 : 0;
        xa.scale = at.scale;
        CDX_TRY(cdx_cross_attention_f32(&xa, st));
        CDX_TRY(gemm(st, B.att, d, k.ca_out_w, d, k.ca_out_b, B.h, d, rows, d, d, CDX_ACT_NONE, nullptr, 0, 1, B.h, d));
        CDX_TRY(layernorm(st, B.h, B.y, rows, d, 1e-5f, k.ln3_g, k.ln3_b, nullptr, nullptr, 0, 1, 0));
        CDX_TRY(gemm(st, B.y, d, k.ff1_w, d, k.ff1_b, B.f, 4 * d, rows, 4 * d, d, CDX_ACT_GELU_ERF));
        CDX_TRY(gemm(st, B.f, 4 * d, k.ff2_w, 4 * d, k.ff2_b, B.h, d, rows, d, 4 * d, CDX_ACT_NONE, nullptr, 0, 1, B.h, d));
    }
    CDX_TRY(layernorm(st, B.h, B.y, rows, d, 1e-5f, w->lnf_g, w->lnf_b, nullptr, nullptr, 0, 1, 0));
    CDX_TRY(gemm(st, B.y, d, w->head_w, d, w->head_b, pred, w->act_dim, rows, w->act_dim, d));
    return CDX_OK;
}

// ------------------------------------------------------------------------------------------------
// ChiUNet1d as implicit-GEMM convolutions (global conditioning)
// ------------------------------------------------------------------------------------------------
#define UNET_SPLITK 6
// ------------------------------------------------------------------------------------------------
// LinearAttention core (reference jannerunet.py:84-93): one workgroup per (sample, head); q, k, v of that head in LDS.
// softmax of k runs over the POSITION axis (one thread per channel d), ctx = k v^T is dim_head x dim_head, out = ctx^T q.
// Work is ~4 L dh^2 MACs per head -- a few hundred thousand at the shipped sizes: plain VALU, the GEMMs around it are the MFMA work.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linattn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int heads, int dh,
                                                        float scale) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float* q = sh;                    // [L][dh]
    float* k = q + (size_t)L * dh;    // [L][dh]
    float* v = k + (size_t)L * dh;    // [L][dh]
    float* ctx = v + (size_t)L * dh;  // [dh][dh]
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads, tid = threadIdx.x;
    const int inner = heads * dh, ld = 3 * inner;
    const float* base = qkv + (size_t)b * L * ld + h * dh;
    for (int i = tid; i < L * dh; i += 256) {
        const int n = i / dh, c = i - n * dh;
        q[i] = base[(size_t)n * ld + c] * scale;
        k[i] = base[(size_t)n * ld + inner + c];
        v[i] = base[(size_t)n * ld + 2 * inner + c];
    }
    __syncthreads();
    for (int d = tid; d < dh; d += 256) {                 // softmax over positions, per channel
        float m = -3.0e38f;
        for (int n = 0; n < L; ++n) m = fmaxf(m, k[n * dh + d]);
        float sum = 0.f;
        for (int n = 0; n < L; ++n) { const float e = expf(k[n * dh + d] - m); k[n * dh + d] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int n = 0; n < L; ++n) k[n * dh + d] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < dh * dh; i += 256) {            // ctx[d][e] = sum_n k[n][d] v[n][e]
        const int d = i / dh, e = i - d * dh;
        float acc = 0.f;
        for (int n = 0; n < L; ++n) acc = fmaf(k[n * dh + d], v[n * dh + e], acc);
        ctx[i] = acc;
    }
    __syncthreads();
    float* o = out + (size_t)b * L * inner + h * dh;
    for (int i = tid; i < L * dh; i += 256) {             // out[n][e] = sum_d ctx[d][e] q[n][d]
        const int n = i / dh, e = i - n * dh;
        float acc = 0.f;
        for (int d = 0; d < dh; ++d) acc = fmaf(ctx[d * dh + e], q[n * dh + d], acc);
        o[(size_t)n * inner + e] = acc;
    }
}

struct UNet {                     // one pass over the op list; with dry == true it only measures the workspace
    const cdx_chiunet_weights* w;
    const cdx_sampling* s;
    hipStream_t st;
    Arena a;
    bool dry;
    int nb, bf, b0, rec, n_rec;
    // persistent (prepare) buffers
    float *x, *prev, *xold, *xin, *obs, *e1, *te, *mte, *gobs, *mo, *tfilm, *ofilm, *pred, *splitk, *lobs;
    long long film_total;         // sum of film_out over all blocks (row width of the film tables)
    long long splitk_floats;      // split-K scratch: UNET_SPLITK slices of the largest conv output (rows x model_dim at full length)

    float* take(long long n) { return a.take(n); }

    int conv(const float* x, int lda, const float* wp, const float* bias, int Lin, int Lout, int taps, int cin, int stride, int pad,
             int N, float* out, int ldc, const float* residual, int ldr) {
        if (dry) return CDX_OK;
        cdx_gemm_args g;
        g.A = x; g.W = wp; g.bias = bias; g.gate = nullptr; g.residual = residual; g.table = nullptr; g.C = out;
        g.M = bf * Lout; g.N = N; g.K = taps * cin; g.lda = lda; g.ldw = taps * cin; g.ldc = ldc; g.ldg = 0; g.ldr = ldr;
        g.rows_per_gate = 1; g.table_rows = 0; g.act = CDX_ACT_NONE;
        g.conv_taps = taps; g.conv_cin = cin; g.conv_lin = Lin; g.conv_lout = Lout; g.conv_stride = stride; g.conv_pad = pad;
        g.partial = splitk; g.partial_slices = (splitk && (long long)g.M * N * UNET_SPLITK <= splitk_floats) ? UNET_SPLITK : 0;
        if (g.partial_slices == 0) g.partial = nullptr;
        return cdx_gemm_f32(&g, st);
    }
    // conv whose K-slice sums stay in the split-K scratch, `slot` slices in (a second conv into the same GroupNorm input -- the other
    // half of a channel concat -- appends its slices): the GroupNorm that follows adds them up (gn_slices)
    int conv_partials(const float* x, int lda, const float* wp, int Lin, int Lout, int taps, int cin, int stride, int pad, int N,
                      int slot, int max_slices, int* slices) {
        *slices = 1;
        if (dry) return CDX_OK;
        cdx_gemm_args g;
        g.A = x; g.W = wp; g.bias = nullptr; g.gate = nullptr; g.residual = nullptr; g.table = nullptr; g.C = nullptr;
        g.M = bf * Lout; g.N = N; g.K = taps * cin; g.lda = lda; g.ldw = taps * cin; g.ldc = N; g.ldg = 0; g.ldr = 0;
        g.rows_per_gate = 1; g.table_rows = 0; g.act = CDX_ACT_NONE;
        g.conv_taps = taps; g.conv_cin = cin; g.conv_lin = Lin; g.conv_lout = Lout; g.conv_stride = stride; g.conv_pad = pad;
        const long long mn = (long long)g.M * N;
        g.partial = splitk + (size_t)slot * mn;
        g.partial_slices = (int)((splitk_floats / mn) - slot);
        if (g.partial_slices > max_slices) g.partial_slices = max_slices;
        return cdx_gemm_partials_f32(&g, st, slices);
    }
    bool fold_ok(int L, int C, int G) const {
        static const bool on = [] { const char* e = getenv("CDX_UNET_GN_FOLD"); return !(e && e[0] == '0'); }();      // A/B hook
        return on && splitk != nullptr && cdx_groupnorm_slices_ok(L, C, G) && (long long)bf * L * C * UNET_SPLITK <= splitk_floats;
    }
    int gn_slices(int slices, const float* xbias, float* y, int L, int C, int G, const float* gamma, const float* beta, const float* fa,
                  int ldfa, const float* fb, int ldfb, int film_mode, const float* residual) {
        if (dry) return CDX_OK;
        cdx_gn_args q{};
        q.x = splitk; q.y = y; q.gamma = gamma; q.beta = beta; q.fa = fa; q.fb = fb; q.residual = residual;
        q.B = bf; q.L = L; q.C = C; q.G = G; q.ldx = C; q.ldy = C; q.ldr = C; q.ldfa = ldfa; q.ldfb = ldfb;
        q.fa_row = s->temb_per_sample ? 0 : rec; q.fa_per_sample = s->temb_per_sample; q.film_mode = film_mode;
        q.act = CDX_ACT_MISH; q.eps = 1e-5f;
        return cdx_groupnorm_slices_f32(&q, slices, (long long)bf * L * C, xbias, st);
    }
    int gn(const float* x, float* y, int L, int C, int G, const float* gamma, const float* beta, const float* fa, int ldfa,
           const float* fb, int ldfb, int film_mode, const float* residual) {
        if (dry) return CDX_OK;
        cdx_gn_args q{};          // (zero-initialised: the optional training outputs dgamma_part / dbeta_part must be NULL here)
        q.x = x; q.y = y; q.gamma = gamma; q.beta = beta; q.fa = fa; q.fb = fb; q.residual = residual;
        q.B = bf; q.L = L; q.C = C; q.G = G; q.ldx = C; q.ldy = C; q.ldr = C; q.ldfa = ldfa; q.ldfb = ldfb;
        q.fa_row = s->temb_per_sample ? 0 : rec; q.fa_per_sample = s->temb_per_sample; q.film_mode = film_mode;
        q.act = CDX_ACT_MISH; q.eps = 1e-5f;
        return cdx_groupnorm_f32(&q, st);
    }
    // ChiResidualBlock on [xa | xb] at length L  ->  new buffer (bf * L, cout)
    // `extra` (local conditioning): a (bf * L, cout) tensor added to the block's output -- it rides in the residual 1x1 conv
    int block(const cdx_chiunet_block& k, long long film_off, const float* xa, const float* xb, int L, float** out,
              const float* extra = nullptr) {
        if (extra != nullptr && k.wra == nullptr) { cdx_set_err("local conditioning joins a block with a residual conv only"); return CDX_EINVAL; }
        const int ks = w->kernel_size, pad = ks / 2, co = k.cout;
        const long long n = (long long)bf * L * co;
        float* h1 = take(n);
        float* h2 = take(n);
        float* res = (k.wra != nullptr) ? take(n) : nullptr;
        float* o = take(n);
        // The 3 / 5-tap convs feed a GroupNorm: their K-slice sums (split-K, and the two halves of a channel concat) are added up by the
        // GroupNorm's own load instead of a reduction pass + a round trip of the conv output (fold; CDX_UNET_GN_FOLD=0: the old order).
        const bool fold = fold_ok(L, co, k.groups);
        const int fm = w->cond_predict_scale ? 1 : 2;
        const float *tf = tfilm ? tfilm + film_off : nullptr, *of = ofilm ? ofilm + film_off : nullptr;
        if (fold) {
            int na = 1, nb2 = 0;
            CDX_TRY(conv_partials(xa, k.cin_a, k.w1a, L, L, ks, k.cin_a, 1, pad, co, 0, k.cin_b > 0 ? UNET_SPLITK / 2 : UNET_SPLITK, &na));
            if (k.cin_b > 0) CDX_TRY(conv_partials(xb, k.cin_b, k.w1b, L, L, ks, k.cin_b, 1, pad, co, na, UNET_SPLITK - na, &nb2));
            CDX_TRY(gn_slices(na + nb2, k.b1, h2, L, co, k.groups, k.g1, k.be1, tf, (int)film_total, of, (int)film_total, fm, nullptr));
        } else {
            CDX_TRY(conv(xa, k.cin_a, k.w1a, k.b1, L, L, ks, k.cin_a, 1, pad, co, h1, co, nullptr, 0));
            if (k.cin_b > 0) CDX_TRY(conv(xb, k.cin_b, k.w1b, nullptr, L, L, ks, k.cin_b, 1, pad, co, h1, co, h1, co));
            CDX_TRY(gn(h1, h2, L, co, k.groups, k.g1, k.be1, tf, (int)film_total, of, (int)film_total, fm, nullptr));
        }
        const float* skip = xa;                                                                 // identity skip
        if (k.wra != nullptr) {                           // (before conv2: the 1 x 1 convs may split K through the same scratch)
            CDX_TRY(conv(xa, k.cin_a, k.wra, k.br, L, L, 1, k.cin_a, 1, 0, co, res, co, extra, co));
            if (k.cin_b > 0) CDX_TRY(conv(xb, k.cin_b, k.wrb, nullptr, L, L, 1, k.cin_b, 1, 0, co, res, co, res, co));
            skip = res;
        }
        if (fold) {
            int n2 = 1;
            CDX_TRY(conv_partials(h2, co, k.w2, L, L, ks, co, 1, pad, co, 0, UNET_SPLITK, &n2));
            CDX_TRY(gn_slices(n2, k.b2, o, L, co, k.groups, k.g2, k.be2, nullptr, 0, nullptr, 0, 0, skip));
        } else {
            CDX_TRY(conv(h2, co, k.w2, k.b2, L, L, ks, co, 1, pad, co, h1, co, nullptr, 0));   // h1 is free again
            CDX_TRY(gn(h1, o, L, co, k.groups, k.g2, k.be2, nullptr, 0, nullptr, 0, 0, skip));
        }
        *out = o;
        return CDX_OK;
    }
    // LinearAttention site `ai` on `cur` (bf * L rows of C channels) -> new buffer; NULL table: nothing to do
    int attention(int ai, const float** cur, int L, int C) {
        if (w->attn == nullptr) return CDX_OK;
        const cdx_unet_attn& A = w->attn[ai];
        const int inner = A.heads * A.dim_head;
        const long long rows = (long long)bf * L;
        float* xn = take(rows * C);
        float* qkv = take(rows * 3 * inner);
        float* core = take(rows * inner);
        float* y = take(rows * C);
        if (dry) { *cur = y; return CDX_OK; }
        CDX_TRY(layernorm(st, *cur, xn, (int)rows, C, 1e-5f, A.ln_g, A.ln_b, nullptr, nullptr, 0, 1, 0));
        CDX_TRY(gemm(st, xn, C, A.qkv_w, C, nullptr, qkv, 3 * inner, (int)rows, 3 * inner, C));
        CDX_TRY(cdx_linattn_f32(qkv, core, bf, L, A.heads, A.dim_head, 1.0f / sqrtf((float)A.dim_head), st));
        CDX_TRY(gemm(st, core, inner, A.out_w, inner, A.out_b, y, C, (int)rows, C, inner, CDX_ACT_NONE, nullptr, 0, 1, xn, C));
        *cur = y;
        return CDX_OK;
    }
    int film_out(const cdx_chiunet_block& k) const { return (w->cond_predict_scale ? 2 : 1) * k.cout; }
    int n_blocks() const { return 2 * w->n_levels + 2 + 2 * (w->n_levels - 1); }
    int n_film_blocks() const { return n_blocks() + (w->local_obs_dim > 0 ? 2 : 0); }      // + local_cond_encoder.0 / .1

    // request-invariant tables: embeddings and the two halves of every block's FiLM vector
    int prepare() {
        const int E = w->emb_dim, EH = w->emb_hidden, EO = w->emb_out, trow = s->temb_per_sample ? bf : n_rec;
        const bool has_obs = w->cond_dim > 0;
        film_total = 0;
        for (int i = 0; i < n_film_blocks(); ++i) film_total += film_out(w->blocks[i]);
        lobs = take(w->local_obs_dim > 0 ? (long long)bf * w->Ta * w->local_obs_dim : 0);
        x = take((long long)nb * s->hd); prev = take((long long)nb * s->hd); xold = take((long long)nb * s->hd);
        xin = take((long long)bf * s->hd);
        obs = take(has_obs ? (long long)bf * w->cond_dim : 0);
        e1 = take((long long)trow * EH); te = take((long long)trow * (EO > E ? EO : E)); mte = take((long long)trow * (EO > E ? EO : E));
        gobs = take(has_obs ? (long long)bf * EO : 0); mo = take(has_obs ? (long long)bf * EO : 0);
        tfilm = take((long long)trow * film_total); ofilm = take(has_obs ? (long long)bf * film_total : 0);
        pred = take((long long)bf * s->hd);
        splitk_floats = (long long)UNET_SPLITK * bf * w->Ta * w->model_dim;     // every conv output is <= bf*Ta*model_dim floats
        splitk = take(splitk_floats);
        if (dry) return CDX_OK;
        if (!has_obs) ofilm = nullptr;
        const float* trows = s->temb;                    // per-sample timesteps: rows of this chunk (both CFG halves alike)
        if (s->temb_per_sample) {
            for (int half = 0; half < bf / nb; ++half)
                if (hipMemcpyAsync(mte + (size_t)half * nb * E, s->temb + (size_t)b0 * E, (size_t)nb * E * sizeof(float),
                                   hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
            trows = mte;                                 // staged in mte, consumed by the first GEMM before mte is rewritten
        }
        CDX_TRY(gemm(st, trows, E, w->map0_w, E, w->map0_b, e1, EH, trow, EH, E, CDX_ACT_MISH));
        CDX_TRY(gemm(st, e1, EH, w->map2_w, EH, w->map2_b, te, EO, trow, EO, EH));
        CDX_TRY(cdx_act_f32(te, mte, (long long)trow * EO, CDX_ACT_MISH, st));
        if (has_obs) {
            const long long n = (long long)bf * w->cond_dim;
            const int n_cond_rows = (s->cond == nullptr || s->cfg_mode == 0) ? 0 : nb;
            hipLaunchKernelGGL(obs_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, obs, s->cond, bf, nb, b0,
                               w->cond_dim, n_cond_rows);
            CDX_TRY(hip_ok());
            CDX_TRY(gemm(st, obs, w->cond_dim, w->gce_w, w->cond_dim, w->gce_b, gobs, EO, bf, EO, w->cond_dim));
            CDX_TRY(cdx_act_f32(gobs, mo, (long long)bf * EO, CDX_ACT_MISH, st));
        }
        if (w->local_obs_dim > 0) {                       // observation rows, one per (sample, position); unconditional samples: zeros
            const int width = w->Ta * w->local_obs_dim;
            const long long n = (long long)bf * width;
            const int n_cond_rows = (s->cond == nullptr || s->cfg_mode == 0) ? 0 : nb;
            hipLaunchKernelGGL(obs_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lobs, s->cond, bf, nb, b0,
                               width, n_cond_rows);
            CDX_TRY(hip_ok());
        }
        long long off = 0;
        for (int i = 0; i < n_film_blocks(); ++i) {
            const cdx_chiunet_block& k = w->blocks[i];
            const int fo = film_out(k);
            // without an observation half the block's bias rides with the time half
            CDX_TRY(gemm(st, mte, EO, k.film_w, w->film_ld, has_obs ? nullptr : k.film_b, tfilm + off, (int)film_total, trow, fo, EO));
            if (has_obs)
                CDX_TRY(gemm(st, mo, EO, k.film_w + EO, w->film_ld, k.film_b, ofilm + off, (int)film_total, bf, fo, EO));
            off += fo;
        }
        return CDX_OK;
    }

    int forward(const float* xrows, float* out_pred) {
        const int nl = w->n_levels, Ta = w->Ta;
        const float* skips[8];
        const float* cur = xrows;
        long long foff = 0;
        int bi = 0, L = Ta;
        float* o = nullptr;
        const float *hl0 = nullptr, *hl1 = nullptr;      // local conditioning (reference chiunet.py:153-160)
        if (w->local_obs_dim > 0) {
            long long lf = 0;
            for (int i = 0; i < n_blocks(); ++i) lf += film_out(w->blocks[i]);
            const cdx_chiunet_block &l1 = w->blocks[n_blocks()], &l2 = w->blocks[n_blocks() + 1];
            float *t1 = nullptr, *t2 = nullptr;
            CDX_TRY(block(l1, lf, lobs, nullptr, Ta, &t1));
            CDX_TRY(block(l2, lf + film_out(l1), lobs, nullptr, Ta, &t2));
            const int md = w->model_dim;
            float* d = take((long long)bf * (Ta / 2) * md);
            CDX_TRY(conv(t2, md, w->lc_down_w, w->lc_down_b, Ta, Ta / 2, 3, md, 2, 1, md, d, md, nullptr, 0));
            hl0 = t1; hl1 = d;
        }
        for (int k = 0; k < nl; ++k) {
            for (int j = 0; j < 2; ++j) {
                const cdx_chiunet_block& kb = w->blocks[bi];
                CDX_TRY(block(kb, foff, cur, nullptr, L, &o, (k == 0 && j == 0) ? hl0 : nullptr));
                foff += film_out(kb); ++bi; cur = o;
            }
            CDX_TRY(attention(k, &cur, L, w->blocks[bi - 1].cout));
            skips[k] = cur;
            if (k < nl - 1) {
                const int C = w->blocks[bi - 1].cout;
                float* d = take((long long)bf * (L / 2) * C);
                CDX_TRY(conv(cur, C, w->down_w[k], w->down_b[k], L, L / 2, 3, C, 2, 1, C, d, C, nullptr, 0));
                cur = d; L /= 2;
            }
        }
        for (int j = 0; j < 2; ++j) {
            const cdx_chiunet_block& kb = w->blocks[bi];
            CDX_TRY(block(kb, foff, cur, nullptr, L, &o));
            foff += film_out(kb); ++bi; cur = o;
            if (j == 0) CDX_TRY(attention(nl, &cur, L, kb.cout));
        }
        for (int k = 0; k < nl - 1; ++k) {
            const float* skip = skips[nl - 1 - k];
            CDX_TRY(block(w->blocks[bi], foff, cur, skip, L, &o, k == nl - 2 ? hl1 : nullptr));
            foff += film_out(w->blocks[bi]); ++bi; cur = o;
            CDX_TRY(block(w->blocks[bi], foff, cur, nullptr, L, &o));
            foff += film_out(w->blocks[bi]); ++bi; cur = o;
            const int C = w->blocks[bi - 1].cout;
            CDX_TRY(attention(nl + 1 + k, &cur, L, C));
            float* u = take((long long)bf * 2 * L * C);   // row (b, j) of the (bf*L, 2C) view = [out[2j] | out[2j+1]]
            CDX_TRY(conv(cur, C, w->up_w_even[k], w->up_b[k], L, L, 2, C, 1, 1, C, u, 2 * C, nullptr, 0));
            CDX_TRY(conv(cur, C, w->up_w_odd[k], w->up_b[k], L, L, 2, C, 1, 0, C, u + C, 2 * C, nullptr, 0));
            cur = u; L *= 2;
        }
        const int md = w->model_dim, ks = w->kernel_size;
        float* f1 = take((long long)bf * L * md);
        float* f2 = take((long long)bf * L * md);
        CDX_TRY(conv(cur, md, w->fin_w, w->fin_b, L, L, ks, md, 1, ks / 2, md, f1, md, nullptr, 0));
        CDX_TRY(gn(f1, f2, L, md, w->final_groups, w->fin_g, w->fin_be, nullptr, 0, nullptr, 0, 0, nullptr));
        CDX_TRY(conv(f2, md, w->out_w, w->out_b, L, L, 1, md, 1, 0, w->act_dim, out_pred, w->act_dim, nullptr, 0));
        return CDX_OK;
    }
};

int chiunet_check(const cdx_chiunet_weights* w, const cdx_sampling* s) {
    if (!w || !w->blocks || !w->map0_w || !w->map2_w || (w->cond_dim > 0 && !w->gce_w) || !w->fin_w || !w->out_w ||
        (w->n_levels > 1 && (!w->down_w || !w->down_b || !w->up_w_even || !w->up_w_odd || !w->up_b))) {
        cdx_set_err("null pointer in ChiUNet1d weights"); return CDX_EINVAL;
    }
    if (w->n_levels < 1 || w->n_levels > 8 || w->Ta <= 0 || (w->Ta >> (w->n_levels - 1)) < 1 || (w->Ta & (w->Ta - 1)) ||
        w->kernel_size < 1 || !(w->kernel_size & 1) || w->emb_dim <= 0 || w->cond_dim < 0 || w->emb_hidden <= 0 || w->emb_out <= 0 ||
        w->film_ld < w->emb_out) {
        cdx_set_err("ChiUNet1d executor: Ta must be a power of two >= 2^(levels-1), odd kernel size"); return CDX_EINVAL;
    }
    CDX_TRY(check_request(s, "cdx_chiunet_run", 7));
    if (s->hd != w->Ta * w->act_dim || s->emb_dim != w->emb_dim) { cdx_set_err("U-Net request: shape mismatch"); return CDX_EINVAL; }
    if (w->cond_dim > 0 && (!s->cond || s->cond_dim != w->cond_dim || s->cfg_mode == 0)) {
        cdx_set_err("ChiUNet1d request: no condition (the reference requires one)"); return CDX_EINVAL;
    }
    if (w->local_obs_dim > 0) {
        if (w->cond_dim != 0 || !w->lc_down_w || !w->lc_down_b || w->n_levels < 2) { cdx_set_err("ChiUNet1d local conditioning: cond_dim 0, lc_down and >= 2 levels required"); return CDX_EINVAL; }
        if (!s->cond || s->cond_dim != w->Ta * w->local_obs_dim || s->cfg_mode == 0) {
            cdx_set_err("ChiUNet1d request: local conditioning needs cond (batch, Ta * obs_dim)"); return CDX_EINVAL;
        }
        return CDX_OK;
    }
    if (w->cond_dim == 0 && (s->cond || s->cfg_mode != 0)) { cdx_set_err("unconditional U-Net: cond must be NULL, cfg_mode 0"); return CDX_EINVAL; }
    return CDX_OK;
}

long long chiunet_pass(const cdx_chiunet_weights* w, const cdx_sampling* s, hipStream_t st, float* base, bool dry, int* rc) {
    const int chunk = chunk_of(s);
    long long need = 0;
    *rc = CDX_OK;
    for (int b0 = 0; b0 < s->batch; b0 += chunk) {
        UNet u;
        u.w = w; u.s = s; u.st = st; u.a = Arena{base, 0, 0}; u.dry = dry;
        u.nb = s->batch - b0 < chunk ? s->batch - b0 : chunk;
        if (dry) u.nb = chunk;
        u.bf = u.nb * (s->cfg_mode == 2 ? 2 : 1); u.b0 = b0; u.rec = 0; u.n_rec = s->n_steps > 0 ? s->n_steps : 1;
        if ((*rc = u.prepare()) != CDX_OK) return -1;
        const long long mark = u.a.used;
        const size_t off = (size_t)b0 * s->hd, bytes = (size_t)u.nb * s->hd * sizeof(float);
        const int n_fwd = s->n_steps > 0 ? s->n_steps : 1;
        for (int i = 0; i < n_fwd; ++i) {
            u.a.used = mark;                              // activations of a forward are recycled every step
            u.rec = i;
            const float* xsrc = s->n_steps == 0 ? s->x_in + off : u.x;
            if (!dry) {
                if (i == 0 && s->n_steps > 0 && hipMemcpyAsync(u.x, s->x_in + off, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { *rc = hip_ok(); return -1; }
                const float in_scale = (s->n_steps > 0 && s->steps[i].kind >= 5) ? s->steps[i].alpha : 1.0f;     // EDM: F(c_in x, ...)
                for (int half = 0; half < u.bf / u.nb; ++half) {
                    float* xdst = u.xin + (size_t)half * u.nb * s->hd;
                    if (in_scale != 1.0f) {
                        const float* src = xsrc;
                        if ((*rc = scaled_input(st, src, xdst, in_scale, (size_t)u.nb * s->hd)) != CDX_OK) return -1;
                    } else if (hipMemcpyAsync(xdst, xsrc, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { *rc = hip_ok(); return -1; }
                }
            }
            float* dst = (s->n_steps == 0) ? s->x_out + off : u.pred;
            if ((*rc = u.forward(u.xin, dst)) != CDX_OK) return -1;
            if (u.a.used > need) need = u.a.used;
            if (dry) break;
            if (s->n_steps > 0 && (*rc = run_step(st, s, s->steps[i], u.x, u.pred, u.prev, u.xold, u.nb, b0)) != CDX_OK) return -1;
        }
        if (dry) break;
        if (s->n_steps > 0 && hipMemcpyAsync(s->x_out + off, u.x, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { *rc = hip_ok(); return -1; }
    }
    return need;
}

// One helper stream + two events per device for cdx_guided_run's fork/join (created on first use, never destroyed: process lifetime)
struct SideStream {
    hipStream_t stream;
    hipEvent_t fork, join;
    std::mutex* busy;            // held for a whole guided call: one side stream + event pair per device, one user at a time
};
SideStream* side_stream() {
    static const bool enabled = [] { const char* e = getenv("CDX_GUIDED_OVERLAP"); return !(e && e[0] == '0'); }();
    if (!enabled) return nullptr;
    static SideStream slots[64];
    static bool ready[64] = {};
    static bool failed[64] = {};
    static std::mutex init_lock, busy_locks[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> guard(init_lock);       // lazy creation from concurrent callers (threads / streams)
    if (failed[dev]) return nullptr;
    if (!ready[dev]) {
        SideStream s;
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) {
            failed[dev] = true;
            (void)hipGetLastError();
            return nullptr;
        }
        s.busy = &busy_locks[dev];
        slots[dev] = s;
        ready[dev] = true;
    }
    return &slots[dev];
}

// ------------------------------------------------------------------------------------------------
// HalfJannerUNet1d forward + input-gradient backward (classifier guidance)
// ------------------------------------------------------------------------------------------------
__global__ void fill_kernel(float* __restrict__ p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

#define HJ_SPLITK 4
struct HjPass {
    const cdx_hjgrad_weights* w;
    hipStream_t st;
    Arena a;
    bool dry;
    int b;
    float* partial = nullptr;     // split-K scratch: at batch 256 every conv has <= 64 tiles and a long serial K loop
    long long partial_floats = 0;

    int conv(const float* x, int cin, const float* wp, const float* bias, int Lin, int Lout, int taps, int stride, int pad, int N,
             float* out, int ldc, const float* residual) {
        if (dry) return CDX_OK;
        cdx_gemm_args g;
        g.A = x; g.W = wp; g.bias = bias; g.gate = nullptr; g.residual = residual; g.table = nullptr; g.C = out;
        g.M = b * Lout; g.N = N; g.K = taps * cin; g.lda = cin; g.ldw = taps * cin; g.ldc = ldc; g.ldg = 0; g.ldr = N;
        g.rows_per_gate = 1; g.table_rows = 0; g.act = CDX_ACT_NONE;
        g.conv_taps = taps; g.conv_cin = cin; g.conv_lin = Lin; g.conv_lout = Lout; g.conv_stride = stride; g.conv_pad = pad;
        const bool fits = partial && (long long)g.M * N * HJ_SPLITK <= partial_floats && ldc == N;
        g.partial = fits ? partial : nullptr; g.partial_slices = fits ? HJ_SPLITK : 0;
        return cdx_gemm_f32(&g, st);
    }
    int gn(bool backward, const float* x, float* y, int L, int C, int G, const float* gamma, const float* beta, const float* fa,
           const float* res_or_dy) {
        if (dry) return CDX_OK;
        cdx_gn_args q{};          // (zero-initialised: the optional training outputs dgamma_part / dbeta_part must be NULL here)
        q.x = x; q.y = y; q.gamma = gamma; q.beta = beta; q.fa = fa; q.fb = nullptr; q.residual = res_or_dy;
        q.B = b; q.L = L; q.C = C; q.G = G; q.ldx = C; q.ldy = C; q.ldr = C; q.ldfa = C; q.ldfb = 0; q.fa_row = 0;
        q.fa_per_sample = 1; q.film_mode = fa ? 2 : 0; q.act = CDX_ACT_MISH; q.eps = 1e-5f;
        return backward ? cdx_groupnorm_bwd_f32(&q, st) : cdx_groupnorm_f32(&q, st);
    }
    int lin(const float* A, int lda, const float* W, int K, const float* bias, float* C, int M, int N, int act, const float* gate,
            const float* residual) {
        if (dry) return CDX_OK;
        return gemm(st, A, lda, W, K, bias, C, N, M, N, K, act, gate, N, 1, residual, N);
    }

    int run(const float* x, const float* emb0, int emb0_ld, float* logp, float* grad) {
        const int md = w->model_dim, H = w->horizon, D = w->in_dim;
        {
            long long widest = (long long)H * D;           // floats per sample of the largest conv output
            int L = H, bi = 0, di = 0;
            for (int s = 0; s < w->n_stages; ++s) {
                if (w->stage_kind[s] == 0) { const long long n = (long long)L * w->blocks[bi++].cout; widest = n > widest ? n : widest; }
                else { ++di; L = (L - 1) / 2 + 1; }
            }
            partial_floats = (long long)HJ_SPLITK * b * widest;
            partial = a.take(partial_floats);
            if (dry) partial = nullptr;
        }
        float* e1 = a.take((long long)b * 4 * md);
        float* emb = a.take((long long)b * md);
        float* memb = a.take((long long)b * md);
        CDX_TRY(lin(emb0, emb0_ld, w->map0_w, w->emb_dim, w->map0_b, e1, b, 4 * md, CDX_ACT_MISH, nullptr, nullptr));
        CDX_TRY(lin(e1, 4 * md, w->map2_w, 4 * md, w->map2_b, emb, b, md, CDX_ACT_NONE, nullptr, nullptr));
        if (!dry) CDX_TRY(cdx_act_f32(emb, memb, (long long)b * md, CDX_ACT_MISH, st));
        // ---- forward; a1 / a2 of every block are kept for the backward
        const float* cur = x;
        int L = H, bi = 0, di = 0;
        struct Saved { float *a1, *a2; int L; } saved[64];
        if (w->n_stages > 64) { cdx_set_err("cdx_hjgrad_run: too many stages"); return CDX_EINVAL; }
        for (int s = 0; s < w->n_stages; ++s) {
            if (w->stage_kind[s] == 0) {
                const cdx_hj_block& k = w->blocks[bi++];
                const long long n = (long long)b * L * k.cout;
                float* e = a.take((long long)b * k.cout);
                float *a1 = a.take(n), *h1 = a.take(n), *a2 = a.take(n), *o = a.take(n);
                float* res = k.wr ? a.take(n) : nullptr;
                CDX_TRY(lin(memb, md, k.emb_w, md, k.emb_b, e, b, k.cout, CDX_ACT_NONE, nullptr, nullptr));
                CDX_TRY(conv(cur, k.cin, k.w1, k.b1, L, L, k.k, 1, k.k / 2, k.cout, a1, k.cout, nullptr));
                CDX_TRY(gn(false, a1, h1, L, k.cout, k.groups, k.g1, k.be1, e, nullptr));
                CDX_TRY(conv(h1, k.cout, k.w2, k.b2, L, L, k.k, 1, k.k / 2, k.cout, a2, k.cout, nullptr));
                if (k.wr) CDX_TRY(conv(cur, k.cin, k.wr, k.br, L, L, 1, 1, 0, k.cout, res, k.cout, nullptr));
                CDX_TRY(gn(false, a2, o, L, k.cout, k.groups, k.g2, k.be2, nullptr, k.wr ? res : cur));
                saved[s] = Saved{a1, a2, L};
                cur = o;
            } else {
                const cdx_hj_down& k = w->downs[di++];
                const int Lo = (L - 1) / 2 + 1;
                float* o = a.take((long long)b * Lo * k.c);
                CDX_TRY(conv(cur, k.c, k.w, k.b, L, Lo, 3, 2, 1, k.c, o, k.c, nullptr));
                saved[s] = Saved{nullptr, nullptr, L};
                cur = o; L = Lo;
            }
        }
        const int fc_in = w->c_last * w->l_last, fh = w->fc_hidden, od = w->out_dim;
        float *u0 = a.take((long long)b * fh), *u = a.take((long long)b * fh), *hm = a.take((long long)b * fh);
        float *gu = a.take((long long)b * fh), *ones = a.take((long long)b * od), *du = a.take((long long)b * fh);
        float* g = a.take((long long)b * fc_in);
        CDX_TRY(lin(emb, md, w->fc1_we, md, w->fc1_b, u0, b, fh, CDX_ACT_NONE, nullptr, nullptr));
        CDX_TRY(lin(cur, fc_in, w->fc1_wx, fc_in, nullptr, u, b, fh, CDX_ACT_NONE, nullptr, u0));
        if (!dry) {
            CDX_TRY(cdx_act_f32(u, hm, (long long)b * fh, CDX_ACT_MISH, st));
            CDX_TRY(cdx_act_f32(u, gu, (long long)b * fh, CDX_ACT_MISH_GRAD, st));
            const size_t n1 = (size_t)b * od;
            hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, ones, 1.0f, n1);
            CDX_TRY(hip_ok());
        }
        CDX_TRY(lin(hm, fh, w->fc2_w, fh, w->fc2_b, logp, b, od, CDX_ACT_NONE, nullptr, nullptr));
        // ---- backward of logp.sum()
        CDX_TRY(lin(ones, od, w->fc2_w_t, od, nullptr, du, b, fh, CDX_ACT_NONE, gu, nullptr));
        CDX_TRY(lin(du, fh, w->fc1_wx_t, fh, nullptr, g, b, fc_in, CDX_ACT_NONE, nullptr, nullptr));
        const float* gcur = g;                             // (b * L, C) rows of the current gradient
        for (int s = w->n_stages - 1; s >= 0; --s) {
            const int Ls = saved[s].L;
            if (w->stage_kind[s] == 1) {
                const cdx_hj_down& k = w->downs[--di];
                const int Lo = (Ls - 1) / 2 + 1;
                float* full = a.take((long long)b * Ls * k.c);
                if (Ls == 1) {
                    CDX_TRY(conv(gcur, k.c, k.bwd_even, nullptr, 1, 1, 1, 1, 0, k.c, full, k.c, nullptr));
                } else {                                   // row (b, m) of the (b*Lo, 2c) view = [dX[2m] | dX[2m+1]]
                    CDX_TRY(conv(gcur, k.c, k.bwd_even, nullptr, Lo, Lo, 1, 1, 0, k.c, full, 2 * k.c, nullptr));
                    CDX_TRY(conv(gcur, k.c, k.bwd_odd, nullptr, Lo, Lo, 2, 1, 0, k.c, full + k.c, 2 * k.c, nullptr));
                }
                gcur = full;
            } else {
                const cdx_hj_block& k = w->blocks[--bi];
                const long long n = (long long)b * Ls * k.cout;
                float *da2 = a.take(n), *dh1 = a.take(n), *da1 = a.take(n);
                float* dres = k.wr ? a.take((long long)b * Ls * k.cin) : nullptr;
                const bool last = s == 0;
                float* dx = last ? grad : a.take((long long)b * Ls * k.cin);
                CDX_TRY(gn(true, saved[s].a2, da2, Ls, k.cout, k.groups, k.g2, k.be2, nullptr, gcur));
                CDX_TRY(conv(da2, k.cout, k.w2_bwd, nullptr, Ls, Ls, k.k, 1, k.k / 2, k.cout, dh1, k.cout, nullptr));
                CDX_TRY(gn(true, saved[s].a1, da1, Ls, k.cout, k.groups, k.g1, k.be1, nullptr, dh1));
                if (k.wr) CDX_TRY(conv(gcur, k.cout, k.wr_bwd, nullptr, Ls, Ls, 1, 1, 0, k.cin, dres, k.cin, nullptr));
                CDX_TRY(conv(da1, k.cout, k.w1_bwd, nullptr, Ls, Ls, k.k, 1, k.k / 2, k.cin, dx, k.cin, k.wr ? dres : gcur));
                gcur = dx;
            }
        }
        if (w->n_stages > 0 && w->stage_kind[0] != 0) { cdx_set_err("cdx_hjgrad_run: the first stage must be a residual block"); return CDX_EINVAL; }
        return CDX_OK;
    }
};

}  // namespace

extern "C" {

long long cdx_dit1d_workspace_floats(const cdx_dit1d_weights* w, const cdx_sampling* s) {
    if (!w || !s) return -1;
    return dit_layout(w, s, nullptr, nullptr);
}

int cdx_dit1d_run(const cdx_dit1d_weights* w, const cdx_sampling* s, void* hip_stream) {
    CDX_TRY(dit_check(w, s));
    if (s->batch == 0) return CDX_OK;
    DitBuffers B;
    const long long need = dit_layout(w, s, s->workspace, &B);
    if (!s->workspace || s->workspace_floats < need) { cdx_set_err("cdx_dit1d_run: workspace too small"); return CDX_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int chunk = chunk_of(s);
    for (int b0 = 0; b0 < s->batch; b0 += chunk) {
        const int nb = s->batch - b0 < chunk ? s->batch - b0 : chunk;
        const size_t off = (size_t)b0 * s->hd, bytes = (size_t)nb * s->hd * sizeof(float);
        CDX_TRY(dit_prepare(w, s, st, B, nb, b0));
        if (s->n_steps == 0) {
            CDX_TRY(dit_forward(w, s, st, B, s->x_in + off, s->x_out + off, nb, 0));
            continue;
        }
        if (hipMemcpyAsync(B.x, s->x_in + off, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
        for (int i = 0; i < s->n_steps; ++i) {
            CDX_TRY(dit_forward(w, s, st, B, B.x, B.pred, nb, i, s->steps[i].kind >= 5 ? s->steps[i].alpha : 1.0f));
            CDX_TRY(run_step(st, s, s->steps[i], B.x, B.pred, B.prev, B.xold, nb, b0));
        }
        if (hipMemcpyAsync(s->x_out + off, B.x, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
    }
    return CDX_OK;
}

long long cdx_pearcetf_workspace_floats(const cdx_pearcetf_weights* w, const cdx_sampling* s) {
    if (!w || !s) return -1;
    return ptf_layout(w, s, nullptr, nullptr);
}

int cdx_pearcetf_run(const cdx_pearcetf_weights* w, const cdx_sampling* s, void* hip_stream) {
    CDX_TRY(ptf_check(w, s));
    if (s->batch == 0) return CDX_OK;
    PtfBuffers B;
    const long long need = ptf_layout(w, s, s->workspace, &B);
    if (!s->workspace || s->workspace_floats < need) { cdx_set_err("cdx_pearcetf_run: workspace too small"); return CDX_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int chunk = chunk_of(s);
    for (int b0 = 0; b0 < s->batch; b0 += chunk) {
        const int nb = s->batch - b0 < chunk ? s->batch - b0 : chunk;
        const size_t off = (size_t)b0 * s->hd, bytes = (size_t)nb * s->hd * sizeof(float);
        CDX_TRY(ptf_prepare(w, s, st, B, nb, b0));
        if (s->n_steps == 0) {
            CDX_TRY(ptf_forward(w, s, st, B, s->x_in + off, s->x_out + off, nb, 0));
            continue;
        }
        if (hipMemcpyAsync(B.x, s->x_in + off, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
        for (int i = 0; i < s->n_steps; ++i) {
            CDX_TRY(ptf_forward(w, s, st, B, B.x, B.pred, nb, i, s->steps[i].kind >= 5 ? s->steps[i].alpha : 1.0f));
            CDX_TRY(run_step(st, s, s->steps[i], B.x, B.pred, B.prev, B.xold, nb, b0));
        }
        if (hipMemcpyAsync(s->x_out + off, B.x, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
    }
    return CDX_OK;
}

long long cdx_chitf_workspace_floats(const cdx_chitf_weights* w, const cdx_sampling* s) {
    if (!w || !s) return -1;
    return tf_layout(w, s, nullptr, nullptr);
}

int cdx_chitf_run(const cdx_chitf_weights* w, const cdx_sampling* s, void* hip_stream) {
    CDX_TRY(tf_check(w, s));
    if (s->batch == 0) return CDX_OK;
    TfBuffers B;
    const long long need = tf_layout(w, s, s->workspace, &B);
    if (!s->workspace || s->workspace_floats < need) { cdx_set_err("cdx_chitf_run: workspace too small"); return CDX_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int chunk = chunk_of(s);
    for (int b0 = 0; b0 < s->batch; b0 += chunk) {
        const int nb = s->batch - b0 < chunk ? s->batch - b0 : chunk;
        const size_t off = (size_t)b0 * s->hd, bytes = (size_t)nb * s->hd * sizeof(float);
        CDX_TRY(tf_prepare(w, s, st, B, nb, b0));
        if (s->n_steps == 0) {
            CDX_TRY(tf_forward(w, s, st, B, s->x_in + off, s->x_out + off, nb, 0));
            continue;
        }
        if (hipMemcpyAsync(B.x, s->x_in + off, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
        for (int i = 0; i < s->n_steps; ++i) {
            CDX_TRY(tf_forward(w, s, st, B, B.x, B.pred, nb, i, s->steps[i].kind >= 5 ? s->steps[i].alpha : 1.0f));
            CDX_TRY(run_step(st, s, s->steps[i], B.x, B.pred, B.prev, B.xold, nb, b0));
        }
        if (hipMemcpyAsync(s->x_out + off, B.x, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
    }
    return CDX_OK;
}

int cdx_linattn_f32(const float* qkv, float* out, int32_t B, int32_t L, int32_t heads, int32_t dim_head, float scale, void* hip_stream) {
    if (B < 0 || L <= 0 || L > 1024 || heads <= 0 || dim_head <= 0 || dim_head > 64) {
        cdx_set_err("cdx_linattn_f32: L <= 1024 and dim_head <= 64 required"); return CDX_EINVAL;
    }
    if (B == 0) return CDX_OK;
    if (!qkv || !out) { cdx_set_err("cdx_linattn_f32: null pointer"); return CDX_EINVAL; }
    const size_t lds = ((size_t)3 * L * dim_head + (size_t)dim_head * dim_head) * sizeof(float);
    if (lds > 160u * 1024u) { cdx_set_err("cdx_linattn_f32: q, k, v of one head exceed 160 KiB of LDS"); return CDX_ELDS; }
    if (lds > 48u * 1024u &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return hip_ok();
    hipLaunchKernelGGL(linattn_kernel, dim3((unsigned)(B * heads)), dim3(256), lds, reinterpret_cast<hipStream_t>(hip_stream), qkv, out, L,
                       heads, dim_head, scale);
    return hip_ok();
}

long long cdx_chiunet_workspace_floats(const cdx_chiunet_weights* w, const cdx_sampling* s) {
    if (!w || !s || !w->blocks) return -1;
    int rc;
    return chiunet_pass(w, s, nullptr, nullptr, true, &rc);
}

int cdx_chiunet_run(const cdx_chiunet_weights* w, const cdx_sampling* s, void* hip_stream) {
    CDX_TRY(chiunet_check(w, s));
    if (s->batch == 0) return CDX_OK;
    int rc;
    const long long need = chiunet_pass(w, s, nullptr, nullptr, true, &rc);
    if (!s->workspace || s->workspace_floats < need) { cdx_set_err("cdx_chiunet_run: workspace too small"); return CDX_EINVAL; }
    chiunet_pass(w, s, reinterpret_cast<hipStream_t>(hip_stream), s->workspace, false, &rc);
    return rc;
}

long long cdx_hjgrad_workspace_floats(const cdx_hjgrad_weights* w, int32_t batch) {
    if (!w || !w->stage_kind || !w->blocks || batch < 0) return -1;
    HjPass p{w, nullptr, Arena{nullptr, 0, 0}, true, batch};
    if (p.run(nullptr, nullptr, 0, nullptr, nullptr) != CDX_OK) return -1;
    return p.a.used;
}

int cdx_hjgrad_run(const cdx_hjgrad_weights* w, const float* x, const float* emb0, int32_t emb0_ld, int32_t batch, float* logp,
                   float* grad, float* workspace, long long workspace_floats, void* hip_stream) {
    if (!w || !w->stage_kind || !w->blocks || !w->map0_w || !w->fc1_wx || !w->fc2_w) { cdx_set_err("cdx_hjgrad_run: null pointer in weights"); return CDX_EINVAL; }
    if (batch < 0 || w->horizon <= 0 || w->in_dim <= 0) { cdx_set_err("cdx_hjgrad_run: bad shape"); return CDX_EINVAL; }
    if (batch == 0) return CDX_OK;
    if (!x || !emb0 || !logp || !grad) { cdx_set_err("cdx_hjgrad_run: null tensor"); return CDX_EINVAL; }
    const long long need = cdx_hjgrad_workspace_floats(w, batch);
    if (need < 0) return CDX_EINVAL;
    if (!workspace || workspace_floats < need) { cdx_set_err("cdx_hjgrad_run: workspace too small"); return CDX_EINVAL; }
    if (emb0_ld != 0 && emb0_ld < w->emb_dim) { cdx_set_err("cdx_hjgrad_run: emb0_ld must be 0 (shared row) or >= emb_dim"); return CDX_EINVAL; }
    HjPass p{w, reinterpret_cast<hipStream_t>(hip_stream), Arena{workspace, 0, 0}, false, batch};
    return p.run(x, emb0, emb0_ld, logp, grad);
}

namespace {
// forward-mode request of the implicit-GEMM U-Net executor for one guided step (row `step` of the denoiser's timestep table)
cdx_sampling guided_gemm_request(const cdx_guided_launch* g, int step, const float* x, float* pred, float* ws, long long ws_floats) {
    cdx_sampling S{};
    S.batch = g->batch; S.hd = g->hd; S.emb_dim = g->denoiser_emb_dim; S.cond_dim = 0;
    S.temb = g->temb ? g->temb + (size_t)step * g->denoiser_emb_dim : nullptr;
    S.n_steps = 0; S.temb_per_sample = 0; S.cfg_mode = 0;
    S.x_in = x; S.x_out = pred; S.workspace = ws; S.workspace_floats = ws_floats; S.chunk = g->denoiser_chunk;
    return S;
}
}  // namespace

long long cdx_guided_workspace_floats(const cdx_guided_launch* g) {
    if (!g || !g->classifier || g->batch < 0 || g->hd <= 0) return -1;
    const long long clf = cdx_hjgrad_workspace_floats(g->classifier, g->batch);
    if (clf < 0) return -1;
    const long long state = ((long long)g->batch * g->hd + 63) & ~63LL;
    long long den = 0;
    if (g->denoiser_gemm) {
        const cdx_sampling S = guided_gemm_request(g, 0, nullptr, nullptr, nullptr, 0);
        int rc;
        den = chiunet_pass(g->denoiser_gemm, &S, nullptr, nullptr, true, &rc);
        if (den < 0) return -1;
        den = (den + 63) & ~63LL;
    }
    return clf + den + 6 * state + (((long long)g->batch * g->classifier->out_dim + 63) & ~63LL);     // x, pred, prev, grad, x_old, c_in x, logp
}

int cdx_guided_run(const cdx_guided_launch* g, void* hip_stream) {
    if (!g || (!g->denoiser && !g->denoiser_gemm) || !g->classifier || !g->steps || !g->cg_scale || (g->denoiser_gemm && !g->temb) || !g->clf_emb0 ||
        !g->x_in || !g->x_out) {
        cdx_set_err("cdx_guided_run: null pointer"); return CDX_EINVAL;
    }
    if (g->denoiser && g->denoiser_gemm) { cdx_set_err("cdx_guided_run: give the denoiser as a program launch OR as GEMM-executor weights"); return CDX_EINVAL; }
    if (g->n_steps <= 0 || g->batch < 0 || g->classifier->horizon * g->classifier->in_dim != g->hd) {
        cdx_set_err("cdx_guided_run: bad shape"); return CDX_EINVAL;
    }
    if (g->denoiser && (g->hd != g->denoiser->horizon * g->denoiser->dim || g->denoiser->n_steps != 0 || g->denoiser->n_pass == 2 ||
                        g->denoiser->mlp != 0 || g->denoiser->compact != 0 || !g->denoiser->emb)) {
        cdx_set_err("cdx_guided_run: the denoiser must be a forward-mode U-Net launch matching the classifier's (horizon, dim)"); return CDX_EINVAL;
    }
    if (g->denoiser_gemm) {
        if (g->denoiser_emb_dim <= 0 || g->denoiser_chunk < 0) { cdx_set_err("cdx_guided_run: GEMM denoiser needs emb_dim > 0, chunk >= 0"); return CDX_EINVAL; }
        const cdx_sampling S = guided_gemm_request(g, 0, g->x_in, g->x_out, nullptr, 0);
        CDX_TRY(chiunet_check(g->denoiser_gemm, &S));
    }
    if (g->fix_mask && !g->prior) { cdx_set_err("cdx_guided_run: fix_mask given without prior"); return CDX_EINVAL; }
    // step kinds 0-2, or an all-EDM plan (kinds 5 / 6: classifier guidance under ContinuousEDM, reference newedm.py:217-284 -- the
    // prediction D = c_skip x + c_out F is shifted by w sigma^2 grad, i.e. F by cg_scale = w sigma^2 / c_out; the network sees c_in x,
    // the classifier x itself)
    const bool edm = g->steps[0].kind >= 5;
    for (int i = 0; i < g->n_steps; ++i) {
        const int kd = g->steps[i].kind;
        if (kd < 0 || (kd > 2 && kd != 5 && kd != 6) || (kd >= 5) != edm || (g->steps[i].noise_idx >= 0 && !g->noise)) {
            cdx_set_err("cdx_guided_run: step kinds 0-2 or an all-EDM plan (5 / 6); stochastic steps need the noise tensor"); return CDX_EINVAL;
        }
    }
    if (g->batch == 0) return CDX_OK;
    const long long need = cdx_guided_workspace_floats(g);
    if (!g->workspace || g->workspace_floats < need) { cdx_set_err("cdx_guided_run: workspace too small"); return CDX_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    Arena a{g->workspace, 0, 0};
    const long long n = (long long)g->batch * g->hd;
    float *x = a.take(n), *pred = a.take(n), *prev = a.take(n), *grad = a.take(n), *xold = a.take(n), *xs = a.take(n);
    float* logp = a.take((long long)g->batch * g->classifier->out_dim);
    float* den_ws = nullptr;
    long long den_floats = 0;
    if (g->denoiser_gemm) {
        const cdx_sampling S = guided_gemm_request(g, 0, nullptr, nullptr, nullptr, 0);
        int rc0;
        den_floats = chiunet_pass(g->denoiser_gemm, &S, nullptr, nullptr, true, &rc0);
        den_ws = a.take(den_floats);
    }
    float* clf_ws = g->workspace + a.used;
    const long long clf_floats = g->workspace_floats - a.used;
    if (hipMemcpyAsync(x, g->x_in, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
    // The denoiser forward and the classifier forward+backward of a step both read x_t and are independent: the fused U-Net kernel
    // (one workgroup per CU, ~121 KB of LDS) runs on a side stream while the classifier's ~80 small GEMM launches go down the
    // caller's stream -- a GEMM workgroup (34 KB LDS, 4 waves) fits next to the U-Net workgroup on every CU.  Fork/join with two
    // events per step; all side-stream work is joined before the call returns.  CDX_GUIDED_OVERLAP=0 serialises (A/B hook).
    SideStream* side = g->denoiser_gemm ? nullptr : side_stream();      // (the GEMM denoiser's launches stay on the caller's stream)
    // Two guided calls on one device (other threads, other caller streams) would race on the shared fork/join events: the second
    // one waits here.  Every exit after a fork drains the side stream first, so that the caller may free cond / temb / workspace
    // as soon as it has synchronised with ITS stream, error or not.
    std::unique_lock<std::mutex> busy;
    if (side) busy = std::unique_lock<std::mutex>(*side->busy);
    auto bail = [&](int rc) {
        if (side) (void)hipStreamSynchronize(side->stream);
        return rc;
    };
#define CDX_TRY_SIDE(expr) do { const int rc_ = (expr); if (rc_ != CDX_OK) return bail(rc_); } while (0)
    for (int i = 0; i < g->n_steps; ++i) {
        const float* xnet = x;                          // what the denoiser sees: x_t, or c_in x_t under EDM
        if (edm) CDX_TRY(scaled_input(st, xnet, xs, g->steps[i].alpha, (size_t)n));
        if (g->denoiser_gemm) {
            const cdx_sampling S = guided_gemm_request(g, i, xnet, pred, den_ws, den_floats);
            int rcg;
            chiunet_pass(g->denoiser_gemm, &S, st, den_ws, false, &rcg);
            CDX_TRY(rcg);
        }
        cdx_unet2_launch L;
        if (g->denoiser) {
            // forward mode of the program kernel: FiLM row(s) of step i -- one row, or one per trajectory (conditional denoisers)
            L = *g->denoiser;
            L.n_steps = 0; L.steps = nullptr; L.batch = g->batch; L.traj_first = 0; L.traj_count = g->batch;
            L.emb = g->denoiser->emb + (size_t)i * (L.emb_per_traj ? g->batch : 1) * L.emb_ld; L.x_in = xnet; L.x_out = pred;
        }
        if (g->denoiser_gemm) {
        } else if (side) {
            if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->stream, side->fork, 0) != hipSuccess) return bail(hip_ok());
            CDX_TRY_SIDE(cdx_unet2_run(&L, side->stream));
            if (hipEventRecord(side->join, side->stream) != hipSuccess) return bail(hip_ok());
        } else {
            CDX_TRY(cdx_unet2_run(&L, hip_stream));
        }
        const int clf_rc = cdx_hjgrad_run(g->classifier, x, g->clf_emb0 + (size_t)i * g->classifier->emb_dim, 0, g->batch, logp, grad,
                                          clf_ws, clf_floats, hip_stream);
        if (side && hipStreamWaitEvent(st, side->join, 0) != hipSuccess) return bail(hip_ok());      // join even when the classifier failed
        CDX_TRY_SIDE(clf_rc);
        StepArgs sa;
        sa.x = x; sa.pred = pred; sa.prev = prev; sa.xold = xold; sa.prior = g->prior; sa.fix_mask = g->fix_mask;
        sa.noise = g->noise; sa.x_min = g->x_min; sa.x_max = g->x_max; sa.st = g->steps[i]; sa.nb = g->batch; sa.hd = g->hd;
        sa.b0 = 0; sa.batch = g->batch; sa.predict_noise = g->predict_noise; sa.cfg_mode = 0; sa.cfg_w = 0.f;
        sa.grad = grad; sa.cg_scale = g->cg_scale[i];
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(solver_step_kernel, dim3(blocks), dim3(256), 0, st, sa);
        CDX_TRY_SIDE(hip_ok());
    }
#undef CDX_TRY_SIDE
    if (hipMemcpyAsync(g->x_out, x, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
    return CDX_OK;
}

long long cdx_resmlp_workspace_floats(const cdx_resmlp_weights* w, const cdx_sampling* s) {
    if (!w || !s) return -1;
    return mlp_layout(w, s, nullptr, nullptr);
}

int cdx_resmlp_run(const cdx_resmlp_weights* w, const cdx_sampling* s, void* hip_stream) {
    CDX_TRY(mlp_check(w, s));
    if (s->batch == 0) return CDX_OK;
    MlpBuffers B;
    const long long need = mlp_layout(w, s, s->workspace, &B);
    if (!s->workspace || s->workspace_floats < need) { cdx_set_err("cdx_resmlp_run: workspace too small"); return CDX_EINVAL; }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const int chunk = chunk_of(s);
    for (int b0 = 0; b0 < s->batch; b0 += chunk) {
        const int nb = s->batch - b0 < chunk ? s->batch - b0 : chunk;
        const size_t off = (size_t)b0 * s->hd, bytes = (size_t)nb * s->hd * sizeof(float);
        if (s->n_steps == 0) {
            CDX_TRY(mlp_forward(w, s, st, B, s->x_in + off, s->x_out + off, nb, b0, 0, 1.0f));
            continue;
        }
        if (hipMemcpyAsync(B.x, s->x_in + off, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
        for (int i = 0; i < s->n_steps; ++i) {
            const cdx_step& rec = s->steps[i];
            CDX_TRY(mlp_forward(w, s, st, B, B.x, B.pred, nb, b0, i, rec.kind >= 5 ? rec.alpha : 1.0f));
            CDX_TRY(run_step(st, s, rec, B.x, B.pred, B.prev, B.xold, nb, b0));
        }
        if (hipMemcpyAsync(s->x_out + off, B.x, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return hip_ok();
    }
    return CDX_OK;
}

}  // extern "C"
