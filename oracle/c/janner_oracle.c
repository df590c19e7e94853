/* janner_oracle.c -- plain-C restatement of the sampling hot path.  TEST INFRASTRUCTURE (see oracle/__init__.py):
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * No ATen, no BLAS: straight loops in fp32 with double accumulation for the reductions, so it is independent of both
 * PyTorch's CPU kernels and the HIP kernel.  Follows, with citations:
 *   JannerUNet1d.forward      reference cleandiffuser/nn_diffusion/jannerunet.py:154-201
 *   ResidualBlock.forward     :66-69        Downsample1d :24   Upsample1d (ConvTranspose1d k4 s2 p1) :33
 *   GroupNorm1d               reference cleandiffuser/utils/building_blocks.py:60-76  (groups = min(8, C/4), eps 1e-5)
 *   Mish                      x * tanh(softplus(x)), softplus threshold 20 (ATen)
 *   solver step               reference cleandiffuser/diffusion/diffusionsde.py:539-592 in the three affine forms of
 *                             include/cdx.h (the host freezes the scalars exactly as engine/plan.py does)
 * Parameters arrive as ONE flat fp32 buffer in execution order (map_emb, downs, mid blocks, ups, final_conv); the walker
 * below consumes them with a cursor, so a layout mismatch shows up as a wrong answer in tests/test_oracle_ports.py,
 * which pins this file against the fixtures produced by the real reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const float* p; } cur_t;
static const float* take(cur_t* c, size_t n) { const float* r = c->p; c->p += n; return r; }

static float mishf(float x) {
    double sp = x > 20.0f ? (double)x : log1p(exp((double)x));
    return (float)((double)x * tanh(sp));
}

/* y[co][l] = b[co] + sum_ci sum_k w[co][ci][k] x[ci][l*stride + k - pad] */
static void conv1d(const float* x, int ci_n, int len, const float* w, const float* b, int co_n, int k, int stride,
                   int pad, float* y, int len_out) {
    for (int co = 0; co < co_n; ++co)
        for (int l = 0; l < len_out; ++l) {
            double acc = b[co];
            for (int ci = 0; ci < ci_n; ++ci)
                for (int t = 0; t < k; ++t) {
                    int q = l * stride + t - pad;
                    if (q >= 0 && q < len) acc += (double)w[((size_t)co * ci_n + ci) * k + t] * x[(size_t)ci * len + q];
                }
            y[(size_t)co * len_out + l] = (float)acc;
        }
}

/* ConvTranspose1d(k=4, stride=2, pad=1): weight layout (C_in, C_out, k); y[co][2q - 1 + t] += w[ci][co][t] x[ci][q] */
static void conv_transpose1d(const float* x, int ci_n, int len, const float* w, const float* b, int co_n, float* y) {
    const int k = 4, len_out = 2 * len;
    for (int co = 0; co < co_n; ++co)
        for (int p = 0; p < len_out; ++p) {
            double acc = b[co];
            for (int t = 0; t < k; ++t) {
                int num = p + 1 - t;
                if (num < 0 || (num & 1)) continue;
                int q = num / 2;
                if (q >= len) continue;
                for (int ci = 0; ci < ci_n; ++ci)
                    acc += (double)w[((size_t)ci * co_n + co) * k + t] * x[(size_t)ci * len + q];
            }
            y[(size_t)co * len_out + p] = (float)acc;
        }
}

static void group_norm_mish(float* x, int c_n, int len, const float* gamma, const float* beta) {
    int groups = c_n / 4 < 8 ? c_n / 4 : 8, cg = c_n / groups;
    for (int g = 0; g < groups; ++g) {
        double s = 0, s2 = 0;
        int n = cg * len;
        for (int i = 0; i < n; ++i) s += x[(size_t)g * n + i];
        double mean = s / n;
        for (int i = 0; i < n; ++i) { double d = x[(size_t)g * n + i] - mean; s2 += d * d; }
        double rstd = 1.0 / sqrt(s2 / n + 1e-5);
        for (int c = 0; c < cg; ++c)
            for (int l = 0; l < len; ++l) {
                size_t i = ((size_t)g * cg + c) * len + l;
                x[i] = mishf((float)((x[i] - mean) * rstd * gamma[g * cg + c] + beta[g * cg + c]));
            }
    }
}

static void linear(const float* x, int n_in, const float* w, const float* b, int n_out, float* y) {
    for (int o = 0; o < n_out; ++o) {
        double acc = b[o];
        for (int i = 0; i < n_in; ++i) acc += (double)w[(size_t)o * n_in + i] * x[i];
        y[o] = (float)acc;
    }
}

/* ResidualBlock: out = CNA2(CNA1(x) + Linear(Mish(emb))[:, None]) + skip(x).  Returns a malloc'ed (c_out, len). */
static float* resblock(cur_t* c, const float* x, int c_in, int c_out, int len, int k, const float* emb, int md) {
    const float* w1 = take(c, (size_t)c_out * c_in * k); const float* b1 = take(c, c_out);
    const float* g1 = take(c, c_out); const float* be1 = take(c, c_out);
    const float* w2 = take(c, (size_t)c_out * c_out * k); const float* b2 = take(c, c_out);
    const float* g2 = take(c, c_out); const float* be2 = take(c, c_out);
    const float* we = take(c, (size_t)c_out * md); const float* bee = take(c, c_out);
    float* h = malloc(sizeof(float) * c_out * len);
    float* out = malloc(sizeof(float) * c_out * len);
    float* e = malloc(sizeof(float) * c_out);
    float* me = malloc(sizeof(float) * md);
    conv1d(x, c_in, len, w1, b1, c_out, k, 1, k / 2, h, len);
    group_norm_mish(h, c_out, len, g1, be1);
    for (int i = 0; i < md; ++i) me[i] = mishf(emb[i]);
    linear(me, md, we, bee, c_out, e);
    for (int co = 0; co < c_out; ++co) for (int l = 0; l < len; ++l) h[(size_t)co * len + l] += e[co];
    conv1d(h, c_out, len, w2, b2, c_out, k, 1, k / 2, out, len);
    group_norm_mish(out, c_out, len, g2, be2);
    if (c_in != c_out) {
        const float* wr = take(c, (size_t)c_out * c_in); const float* br = take(c, c_out);
        conv1d(x, c_in, len, wr, br, c_out, 1, 1, 0, h, len);
        for (size_t i = 0; i < (size_t)c_out * len; ++i) out[i] += h[i];
    } else {
        for (size_t i = 0; i < (size_t)c_out * len; ++i) out[i] += x[i];
    }
    free(h); free(e); free(me);
    return out;
}

/* One forward for ONE trajectory.  x, y: (horizon, in_dim) row-major; temb (+cond) already summed by the caller. */
void cdx_oracle_janner_forward(const float* params, const float* x, const float* temb, float* y, int horizon,
                               int in_dim, int model_dim, int emb_dim, int kernel_size, int n_levels,
                               const int* dim_mult) {
    cur_t c = { params };
    int dims[16]; dims[0] = in_dim;
    int m = model_dim;
    for (int i = 0; i < n_levels; ++i) { m = (i == 0 ? model_dim : dims[i]) * dim_mult[i]; dims[i + 1] = m; }
    /* map_emb: Linear -> Mish -> Linear */
    const float* w0 = take(&c, (size_t)model_dim * 4 * emb_dim); const float* b0 = take(&c, model_dim * 4);
    const float* w2 = take(&c, (size_t)model_dim * model_dim * 4); const float* b2 = take(&c, model_dim);
    float hid[1024], emb[256];
    linear(temb, emb_dim, w0, b0, model_dim * 4, hid);
    for (int i = 0; i < model_dim * 4; ++i) hid[i] = mishf(hid[i]);
    linear(hid, model_dim * 4, w2, b2, model_dim, emb);
    /* (horizon, in_dim) -> (in_dim, horizon) */
    int len = horizon;
    float* cur = malloc(sizeof(float) * in_dim * len);
    for (int l = 0; l < len; ++l) for (int d = 0; d < in_dim; ++d) cur[(size_t)d * len + l] = x[(size_t)l * in_dim + d];
    float* skips[16]; int skip_len[16];
    for (int i = 0; i < n_levels; ++i) {
        float* a = resblock(&c, cur, dims[i], dims[i + 1], len, kernel_size, emb, model_dim);
        free(cur);
        cur = resblock(&c, a, dims[i + 1], dims[i + 1], len, kernel_size, emb, model_dim);
        free(a);
        skips[i] = malloc(sizeof(float) * dims[i + 1] * len);
        memcpy(skips[i], cur, sizeof(float) * dims[i + 1] * len);
        skip_len[i] = len;
        if (i < n_levels - 1) {
            const float* wd = take(&c, (size_t)dims[i + 1] * dims[i + 1] * 3); const float* bd = take(&c, dims[i + 1]);
            float* dn = malloc(sizeof(float) * dims[i + 1] * (len / 2));
            conv1d(cur, dims[i + 1], len, wd, bd, dims[i + 1], 3, 2, 1, dn, len / 2);
            free(cur); cur = dn; len /= 2;
        }
    }
    int top = dims[n_levels];
    { float* a = resblock(&c, cur, top, top, len, kernel_size, emb, model_dim); free(cur);
      cur = resblock(&c, a, top, top, len, kernel_size, emb, model_dim); free(a); }
    int cc = top;
    for (int u = 0; u < n_levels - 1; ++u) {
        int lvl = n_levels - 1 - u;                 /* consumes skip `lvl`, produces dims[lvl] channels */
        int c_skip = dims[lvl + 1], c_out = dims[lvl];
        float* cat = malloc(sizeof(float) * (cc + c_skip) * len);
        memcpy(cat, cur, sizeof(float) * cc * len);
        memcpy(cat + (size_t)cc * len, skips[lvl], sizeof(float) * c_skip * len);
        free(cur);
        float* a = resblock(&c, cat, cc + c_skip, c_out, len, kernel_size, emb, model_dim); free(cat);
        cur = resblock(&c, a, c_out, c_out, len, kernel_size, emb, model_dim); free(a);
        const float* wu = take(&c, (size_t)c_out * c_out * 4); const float* bu = take(&c, c_out);
        float* up = malloc(sizeof(float) * c_out * len * 2);
        conv_transpose1d(cur, c_out, len, wu, bu, c_out, up);
        free(cur); cur = up; len *= 2; cc = c_out;
    }
    for (int i = 0; i < n_levels; ++i) free(skips[i]);
    (void)skip_len;
    /* final_conv: Conv(k5) -> GN -> Mish -> Conv(1x1) */
    const float* wf = take(&c, (size_t)model_dim * model_dim * 5); const float* bf = take(&c, model_dim);
    const float* gf = take(&c, model_dim); const float* bef = take(&c, model_dim);
    const float* wo = take(&c, (size_t)in_dim * model_dim); const float* bo = take(&c, in_dim);
    float* t = malloc(sizeof(float) * model_dim * len);
    conv1d(cur, cc, len, wf, bf, model_dim, 5, 1, 2, t, len);
    group_norm_mish(t, model_dim, len, gf, bef);
    float* o = malloc(sizeof(float) * in_dim * len);
    conv1d(t, model_dim, len, wo, bo, in_dim, 1, 1, 0, o, len);
    for (int l = 0; l < len; ++l) for (int d = 0; d < in_dim; ++d) y[(size_t)l * in_dim + d] = o[(size_t)d * len + l];
    free(t); free(o); free(cur);
}

/* Whole x-prediction / eps-prediction sampling loop for B trajectories with the cdx_step affine forms.
 * steps: [n_steps][12] floats = {kind, vsel, noise_idx, push, alpha, sigma, k0..k4, pad} (ints stored as floats). */
void cdx_oracle_janner_sample(const float* params, float* x /* in: x_T, out: x_0; (B,H,D) */, const float* prior,
                              const float* fix_mask, const float* noise, const float* temb /* [n_steps][emb_dim] */,
                              const float* steps, int n_steps, int predict_noise, int batch, int horizon, int in_dim,
                              int model_dim, int emb_dim, int kernel_size, int n_levels, const int* dim_mult) {
    const int hd = horizon * in_dim;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < batch; ++b) {
        float* xb = x + (size_t)b * hd;
        float* pred = malloc(sizeof(float) * hd);
        float* prev = calloc(hd, sizeof(float));
        for (int s = 0; s < n_steps; ++s) {
            const float* st = steps + (size_t)s * 12;
            const int kind = (int)st[0], vsel = (int)st[1], nidx = (int)st[2], push = (int)st[3];
            const float al = st[4], sg = st[5], k0 = st[6], k1 = st[7], k2 = st[8], k3 = st[9], k4 = st[10];
            cdx_oracle_janner_forward(params, xb, temb + (size_t)s * emb_dim, pred, horizon, in_dim, model_dim, emb_dim,
                                      kernel_size, n_levels, dim_mult);
            for (int e = 0; e < hd; ++e) {
                const float xv = xb[e], p = pred[e];
                float eps, xth;
                if (predict_noise) { eps = p; xth = (xv - sg * p) / al; } else { xth = p; eps = (xv - al * p) / sg; }
                float xn;
                if (kind == 0) {
                    xn = k0 * (xv - k1 * eps) + k2 * eps;
                    if (nidx >= 0) xn += k3 * noise[((size_t)nidx * batch + b) * hd + e];
                } else if (kind == 1) {
                    xn = k0 * ((xv - k1 * eps) / k2) + k3 * eps;
                } else {
                    float v = vsel == 0 ? eps : xth;
                    if (vsel == 2) v = k3 * xth - k4 * prev[e];
                    xn = k0 * xv - k1 * v;
                    if (nidx >= 0) xn += k2 * noise[((size_t)nidx * batch + b) * hd + e];
                }
                if (fix_mask) { const float m = fix_mask[e]; xn = xn * (1.0f - m) + prior[(size_t)b * hd + e] * m; }
                if (push) prev[e] = xth;
                xb[e] = xn;
            }
        }
        free(pred); free(prev);
    }
}
