/* cdx.h -- C ABI of libcdx.so, the gfx950 (MI355X) executor behind cleandiffuser_amd.
 *
 * The reference (CleanDiffuser) is pure Python on PyTorch and has NO FFI for this path: its boundary is the
 * Python protocol `DiffusionModel.sample()` -> `model["diffusion"](xt, t, cond)` once per denoising step
 * (reference cleandiffuser/diffusion/diffusionsde.py:401-606, loop body :526-594; backbone contract
 * cleandiffuser/nn_diffusion/base_nn_diffusion.py:31-42).  This ABI is what a maintainer would bind *below*
 * that protocol (ctypes stub in INTEGRATION.md): plain pointers and sizes, no torch types, caller-owned memory,
 * no allocation and no synchronisation inside, work enqueued on the caller's HIP stream.
 *
 * Every entry point returns 0 on success or a negative CDX_E* code; cdx_last_error() gives the text.
 */
#ifndef CDX_H_
#define CDX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDX_ABI_VERSION 17

#define CDX_OK 0
#define CDX_EINVAL (-1)   /* bad argument (null pointer, size out of range, misaligned offset) */
#define CDX_ELDS (-2)     /* program needs more LDS than one gfx950 workgroup owns (160 KiB) */
#define CDX_EHIP (-3)     /* HIP runtime error at launch; text in cdx_last_error() */

/* One denoising step = one record.  Replaces the per-step scalar arithmetic of the reference loop
 * (diffusionsde.py:539-589): the host freezes alpha_i, sigma_i and the solver coefficients, the device applies
 *   kind 0 (ddpm)   x <- k0*(x - k1*eps) + k2*eps [+ k3*z]
 *   kind 1 (ddim)   x <- k0*((x - k1*eps)/k2) + k3*eps
 *   kind 2 (linear) x <- k0*x - k1*V [+ k2*z],  V = eps | x_theta | (k3*x_theta - k4*x_theta_prev) | (k3*eps - k4*eps_prev)
 *            with CDX_STEP_MASK_PRED (legacy DPMSolver class, reference diffusion/dpmsolver.py:257-264) the fix-mask is first
 *            applied to the prediction: eps <- eps*(1-m), x_theta <- x_theta*(1-m) + x*m
 *   kind 3/4 (legacy DDPM class, reference diffusion/ddpm.py:153-164,230-241; eps / x0 prediction):
 *            P <- P*(1-mask) [+ x*mask];  x <- k0*(x - k1*P)  |  x <- k0*(k1*x + k2*P);  [+ k3*z]
 *   kind 5/6 (EDM Euler / Heun corrector; reference diffusion/newedm.py:387-401, legacy edm.py:118-160,252-268):
 *            `alpha` carries c_in (the network sees c_in*x), k = (c_skip, c_out, sigma, dt):
 *            D = clip(k0*x + k1*F); s = (x - D)/k2;  kind 5: x' = x - k3*s (push: remember s and x);
 *            kind 6: x' = x_old - k3*(s_old + s)/2.
 *   kind 7 (consistency model, reference diffusion/consistency_model.py:412-427): x' = mask(D) [+ k3*z]  (re-noising for
 *            the next level; the fix-mask is applied BEFORE the noise).   A plan is all-EDM-family (5/6/7) or not at all.
 * followed by the fix-mask blend (diffusionsde.py:592). */
#define CDX_STEP_MASK_PRED 1
typedef struct cdx_step {
    int32_t kind;        /* 0 ddpm, 1 ddim, 2 linear, 3 legacy-ddpm eps, 4 legacy-ddpm x0, 5 edm euler, 6 edm heun, 7 consistency */
    int32_t vsel;        /* kind 2: 0 eps, 1 x_theta, 2 multistep on x_theta, 3 multistep on eps */
    int32_t noise_idx;   /* index into `noise` of this step's N(0,I) draw, or -1 */
    int32_t push;        /* 1: remember x_theta (2: eps) for the next multistep update; EDM: remember slope and state */
    float alpha, sigma;  /* schedule at this step (for eps<->x conversion and clipping) */
    float k[5];
    int32_t flags;       /* CDX_STEP_* bits */
} cdx_step;

/* ABI version of the loaded library (== CDX_ABI_VERSION of the header it was built from). */
int cdx_abi_version(void);

/* What the library needs to know about a device before it may use the launch shapes that assume a whole MI355X (ABI 14): the split /
 * grouped programs of cdx_unet2_run need exactly 256 compute units behind 8 XCDs (one workgroup per CU, 32 per L2) -- a CPX / NPS
 * partition, a CU mask or another gfx9 part must take the ordinary programs.  `xcc_count` is MEASURED: a probe launch of 2048 one-wave
 * workgroups ORs 1 << HW_REG_XCC_ID into *scratch_u32 (caller-owned device word, zeroed by the call); the call synchronises
 * `hip_stream` once (a one-time query, not a hot-path entry).  Returns CDX_OK and fills *out. */
typedef struct cdx_device_props {
    int32_t cu_count;          /* hipDeviceProp_t::multiProcessorCount */
    int32_t xcc_count;         /* distinct HW_REG_XCC_ID values the probe saw */
    int32_t lds_bytes_per_cu;  /* hipDeviceProp_t::maxSharedMemoryPerMultiProcessor */
    int32_t wavefront;         /* hipDeviceProp_t::warpSize */
    char arch[32];             /* hipDeviceProp_t::gcnArchName up to the first ':' ("gfx950") */
} cdx_device_props;
int cdx_device_query(int device, uint32_t* scratch_u32, void* hip_stream, cdx_device_props* out);

/* Text of the last error on the calling thread ("" if none). */
const char* cdx_last_error(void);

/* Activation ids (cdx_gemm_args.act, cdx_act_f32, the op flags of csrc/cdx_ops2.h; mirrored by engine/consts.py). */
#define CDX_ACT_NONE 0
#define CDX_ACT_MISH 1
#define CDX_ACT_GELU_ERF 2
#define CDX_ACT_LEAKY 3
#define CDX_ACT_SILU 4
#define CDX_ACT_RELU 5
#define CDX_ACT_GELU_TANH 6
#define CDX_ACT_MISH_GRAD 7   /* d mish(x) / dx (elementwise map only: classifier-guidance backward) */
#define CDX_ACT_TANH 8        /* critic / inverse-dynamics heads */

/* ------------------------------------------------------------------------------------------------
 * The fused program kernel (csrc/cdx_unet2.hip; program built by engine/program2.py, word layout in csrc/cdx_ops2.h): the whole
 * DiscreteDiffusionSDE / ContinuousDiffusionSDE / ContinuousEDM .sample() loop (reference diffusionsde.py:526-594 over
 * nn_diffusion/jannerunet.py:154-201, chiunet.py:152-192, the MLP denoisers) in ONE launch -- or a single backbone forward
 * (n_steps == 0: x_out <- network(x_in), replaces BaseNNDiffusion.forward).  One workgroup per 1-3 trajectories, activations in LDS,
 * weights streamed as 1-KiB MFMA records: 4 or 8 wave64 per workgroup, `traj_per_wg` trajectories sharing every streamed weight
 * record, the ResidualBlock's 1x1 skip conv fused into its second conv, and the per-block FiLM vectors
 * Linear(Mish(map_emb(map_noise(t)))) read from a per-step table that cdx_unet2_embtab evaluates once per (weights, schedule).
 * ---------------------------------------------------------------------------------------------- */
#define CDX2_OP_WORDS(n_waves) (32 + 8 * (n_waves))   /* header + one inline work item per wave */

/* FiLM table: out[r][:] = W3^T mish(W2^T mish(W0^T temb[r] + b0) + b2) + b3, all weights transposed [n_in][n_out] inside `wblob`
 * at the given float offsets (reference jannerunet.py:135-136 map_emb, :57 emb_mlp of every block, stacked). */
typedef struct cdx_unet2_embtab_args {
    const float* wblob;
    int32_t emb_dim, hidden, md, n_emb;
    int32_t w0, b0, w2, b2, w3, b3;
    const float* temb;        /* device (n_rows, emb_dim): map_noise(t) per step record */
    int32_t n_rows;
    float* out;               /* device (n_rows, out_ld): this network's FiLM vectors go to columns [col0, col0 + n_emb) */
    int32_t out_ld, col0;
    /* optional rows applied to the RAW embedding W2^T mish(...) + b2 (the HalfJannerUNet1d head's share of its first Linear,
     * reference nn_classifier/half_jannerunet.py:62 `cat([x.flatten(1), emb])`): out[r][col4 + o] = W4^T raw + b4; n_raw = 0: none */
    int32_t w4, b4, n_raw, col4;
} cdx_unet2_embtab_args;
int cdx_unet2_embtab(const cdx_unet2_embtab_args* args, void* hip_stream);

typedef struct cdx_unet2_launch {
    const int32_t* ops;        /* device, [n_ops][CDX2_OP_WORDS(n_waves)] (descriptor + first work items) followed by further work items */
    const float* wblob;        /* device, packed parameters */
    int32_t n_ops;
    int32_t traj_floats;       /* LDS floats of one trajectory's region; the workgroup owns traj_per_wg of them */
    int32_t traj_per_wg;       /* 1, 2 or 3 (3: 8-wave compact programs) */
    int32_t n_waves;           /* 4 or 8: wave64 per workgroup; the program (work items, K slices) is compiled for one of them */
    int32_t tune;              /* reserved for scheduling experiments (results never depend on it); 0 */
    int32_t x_off, x_stride, pred_off, pred_stride, prev_off, stage_off;   /* relative to the trajectory region; x/pred: position 0 */
    int32_t batch, horizon, dim;
    /* this launch denoises trajectories [traj_first, traj_first + traj_count) of the batch (all tensors keep their full-batch
     * indexing): lets the host run the bulk of a large batch two trajectories per workgroup and the remainder one per workgroup.
     * Both zero: the whole batch. */
    int32_t traj_first, traj_count;
    const float* emb;          /* device (max(n_steps,1), emb_ld): FiLM table rows, one per step record */
    int32_t emb_ld;
    const cdx_step* steps;     /* device [n_steps], kinds 0-7; NULL with n_steps == 0 (one forward: x_out <- network(x_in)) */
    int32_t n_steps, predict_noise;
    const float* x_in;         /* (batch, horizon, dim) */
    const float* prior;        /* or NULL */
    const float* fix_mask;     /* (horizon, dim) or NULL */
    const float* noise;        /* (n_noise, batch, horizon, dim) or NULL */
    const float* x_min;        /* (horizon, dim) or NULL */
    const float* x_max;
    float* x_out;
    /* init_blend != 0: x_in holds the raw N(0, I) draw z; the kernel forms x_T = (z * x_scale) * (1 - fix_mask) + prior * fix_mask
     * while loading it (reference diffusionsde.py:509-510 `xt = randn_like(prior) * temperature; xt = xt * (1 - mask) + prior * mask`,
     * same roundings), so the host launches nothing but this kernel.  0: x_in is x_T itself. */
    int32_t init_blend;
    float x_scale;
    /* classifier guidance (programs built by engine/program2.py:compile_guided2: the ops after the denoiser's are the classifier's
     * forward and backward-data pass, which leaves d log p / d x_t in the gradient slot): per step the prediction is shifted by
     * cg_scale[step] * gradient before clipping (reference diffusionsde.py:153-173).  NULL: no shift.  with_backward != 0 selects
     * the kernel variant that understands backward ops even without a shift (one forward+backward: gradients only). */
    const float* cg_scale;     /* device [n_steps] or NULL */
    int32_t grad_off, grad_stride, with_backward;
    /* programs compiled with their saved normalised tensors in global memory (so that two trajectories share a workgroup): device
     * scratch of (batch + 1) * ws_floats floats (one spare block), owned by the caller, ordered by the launch stream; ws_floats == 0: none */
    float* ws;
    int32_t ws_floats;
    /* compact programs (engine/program2.py:compile_janner2 / compile_guided2 with compact=True: the LDS plan that lets THREE
     * trajectories share a workgroup, or lets the largest shipped nets fit one workgroup at all): the authoritative state x_t lives
     * in x_out and the multistep memory in the first horizon * dim floats of the trajectory's ws block ((batch + 1) * ws_floats
     * floats; a guided program's saved tensors follow it).  x_out must not alias x_in.  Guided compact programs re-read x_t from
     * x_out for the classifier through a CDX2_KIND2_LOADX op. */
    int32_t compact;
    /* optional profiling: device u64 [n_ops*8 + 2]; workgroup 0 stamps s_memtime at {op start, next-op prefetch issued, after the
     * staging barrier, op end, item record + segment read, first operands landed, MFMAs done, partial tile staged} of the first
     * forward, plus kernel start/end.  NULL = off. */
    unsigned long long* prof;
    /* ---- conditional requests (round 3: what used to need the first program kernel) ----
     * emb_per_traj != 0: `emb` holds one FiLM row per (step, trajectory), row = step * batch + b -- the condition embedding enters
     * JannerUNet1d's time embedding (emb = map_noise(t) + condition, reference jannerunet.py:160-164), so its FiLM vectors are
     * per-sample; with n_steps == 0 this is also how a stand-alone forward gets per-sample timesteps.  The table carries two spare
     * rows at its end (a half-empty last workgroup reads the rows of trajectories past the batch).
     * n_pass == 2: classifier-free-guidance pair (reference diffusionsde.py:175-206): per step one forward with `emb` (conditional)
     * and one with `emb_u` (row = step: the zero-condition embedding), pred = cfg_w * p_cond + (1 - cfg_w) * p_uncond.
     * Step kinds 5/6/7 (EDM Euler / Heun, consistency; a plan is all-EDM or not at all): the network sees c_in * x (c_in = the step's
     * `alpha`), the authoritative state lives in x_out like a compact program's.
     * Any of the three needs ws with ws_floats >= 3 * round4(horizon * dim): [multistep memory or EDM slope | x_old | p_cond]. */
    int32_t emb_per_traj, n_pass;
    const float* emb_u;
    float cfg_w;
    int32_t edm_plan;          /* != 0: the step records are kinds 5-7 */
    /* guided programs: after the last step the classifier's forward ops [logp_first_op, logp_head_op] run once more on the FINAL
     * state with FiLM row `n_steps` of `emb` (timestep 0) and the head writes its scalar to logp_out[b] -- the `log_p` the reference
     * evaluates after the loop (diffusionsde.py:597-601), before the final clip.  NULL: not asked for.
     * With n_steps == 0 (and with_backward != 0, traj_per_wg == 1) the launch is that pass ALONE on x_in -- `BaseClassifier.logp` of a
     * batch (classifier/base.py:62-72; a classifier-only program of engine/program2.py:compile_classifier2): trajectory b reads FiLM
     * row b of `emb` (per-sample timesteps), x_out is not written. */
    float* logp_out;           /* device (batch) or NULL */
    int32_t logp_first_op, logp_head_op;
    /* Batch-tiled MLP programs (engine/program2.py: compile_*_mlp2; reference nn_diffusion/pearcemlp.py, dqlmlp.py, dvinvmlp.py, mlps.py,
     * sfbc_unet.py): mlp != 0 selects the kernel instantiation that decodes their ops -- a "trajectory" is a tile of `horizon` samples
     * of `dim` features (x_in (batch * horizon, dim)), traj_per_wg = 1, n_waves = 8, one table row per step.  `ctx`: the samples'
     * condition features (batch * horizon, C) loaded into the program's context slot at the start of every forward, or NULL (zeros,
     * what the reference substitutes for a missing condition); the second forward of a classifier-free-guidance pair sees zeros. */
    const float* ctx;
    int32_t mlp;
    /* Split programs (engine/program2.py:compile_janner2_split; small batches): split_k = 2 or 4 workgroups per trajectory behind one
     * L2; `ops` holds the members' descriptors back to back (member m's op i = descriptor m * n_ops + i).  After an op that is cut over
     * the members they all-gather its output through `xbuf`: per group two tiles of 2 * xchg_floats floats -- 8-byte {value, tag}
     * granules, tag = the exchange's sequence number, polled until they match.  Sequence numbers start at `xseq0` + 1: the caller keeps
     * them increasing from launch to launch on the same `xbuf` (a granule left by an earlier launch then never matches; zero the buffer
     * when the 32-bit counter would wrap), so nothing has to be cleared per launch.
     * `run_if` (ABI 14; ORDINARY launches only -- split_k == 0): NULL, or a device-visible int32 word: the launch does its work only if
     * *run_if != 0 when it starts, otherwise every workgroup returns at once.  This is the REPAIR launch the host enqueues right behind
     * a split / grouped launch, on the same stream, with the same tensors and run_if = that launch's `xerr`: if a member lost a granule
     * (the polls are bounded) the request is recomputed by the ordinary program before anything downstream of the stream can observe
     * x_out, so a failed exchange never reaches a caller as NaN; if nothing failed the repair costs one empty launch (~10 us).
     * Groups are formed at run time (ABI 13; HIP promises no workgroup -> XCD placement): EVERY split / grouped launch has 256
     * workgroups, one per CU = 32 per XCD, all resident (at most 256 / split_k groups of work; the others compute on zeros); a
     * workgroup draws a ticket from the counter of the XCD it finds itself on (HW_REG_XCC_ID) -- ticket t = member t % split_k of that
     * XCD's group t / split_k -- so the members of a group share an L2 by construction.  Layout of `xbuf` (floats, zeroed once):
     * (256 / split_k) * 4 * xchg_floats of tiles, then 16 x 32 words of ticket counters (one 128-byte line each), then 256 words
     * [group][member] = (xseq0 + 1) << 4 | XCC id of who ended up where (diagnostics).  `xtick0`: tickets drawn per XCD so far on this
     * `xbuf` = 32 x the launches so far (the counters only count up).
     * `xerr`: EIGHT int32 (device memory or pinned host memory): [0] set to 1 if a wait ran out or a ticket did not fit (the polls are
     * bounded; the results are then invalid -- stored as NaN), [1..7] what / workgroup / sequence number / item / XCC id / member /
     * group of the first report.  0 / NULL: an ordinary launch. */
    int32_t split_k, xchg_floats;
    float* xbuf;
    const int32_t* run_if;
    int32_t* xerr;
    uint32_t xseq0;
    uint32_t xtick0;
    /* Grouped programs (engine/program2.py:compile_janner2_group; full batches): split_group != 0 with split_k = 2 or 4 -- the split_k
     * workgroups of a group (behind one L2, as above) own split_k TRAJECTORIES, one each; the ops that are bound by the L2 -> CU weight stream
     * are computed per member for 1/split_k of the output channels of all the group's trajectories (descriptor word W2_XG: CDX2_XG_GOP)
     * and all-gathered through `xbuf` like the cut ops of a split program.  256 workgroups, groups formed from per-XCD tickets as above;
     * trajectory of a workgroup = traj_first + group * split_k + member.  A workgroup that loses a
     * granule sets `xerr` AND stores NaN instead of its result (split programs likewise): a failed exchange never looks like a sample. */
    int32_t split_group;
    /* TEST HOOK (ABI 14): fault = m + 1 makes member m of EVERY group withhold its granules in every exchange of the launch -- a
     * lost granule on purpose (tests/test_gpu_parity.py: sample() must never hand out NaN, the mode must switch itself off).  0: off. */
    int32_t fault;
} cdx_unet2_launch;
int cdx_unet2_run(const cdx_unet2_launch* launch, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Big-batch building blocks (csrc/cdx_gemm.hip): when M = batch x tokens >> 256 the denoiser layers are classic
 * GEMMs.  Tensors are fp32, row-major with explicit leading dimensions, weights in the PyTorch (N, K) layout.
 * ---------------------------------------------------------------------------------------------- */

/* C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) * gate[m / rows_per_gate][n] + residual[m][n] + table[m % table_rows][n]
 * Replaces nn.Linear and its surrounding elementwise ops (reference nn_diffusion/dit.py:31-36,49,118; idqlmlp.py:12-18). */
typedef struct cdx_gemm_args {
    const float* A;        /* (M, K), leading dimension lda */
    const float* W;        /* (N, K), leading dimension ldw */
    const float* bias;     /* (N) or NULL */
    const float* gate;     /* (M / rows_per_gate, ldg) or NULL: multiplies the activated value */
    const float* residual; /* (M, ldr) or NULL: added after the gate */
    const float* table;    /* (table_rows, N) or NULL: added last (positional table) */
    float* C;              /* (M, ldc) */
    int32_t M, N, K, lda, ldw, ldc, ldg, ldr;
    int32_t rows_per_gate, table_rows;
    int32_t act;           /* CDX_ACT_*: 0 none, 1 mish, 2 gelu(erf), 3 leaky, 4 silu, 5 relu, 6 gelu(tanh), 8 tanh */
    /* Implicit-GEMM Conv1d (conv_taps > 0): A is a channel-last activation tensor, rows = samples * conv_lin, row stride lda;
     * output row m = (b, lo) = (m / conv_lout, m % conv_lout); K = conv_taps * conv_cin with k = tap * conv_cin + c;
     * A[m][k] = X[b * conv_lin + lo * conv_stride + tap - conv_pad][c], zero outside [0, conv_lin).  W is (N, conv_taps, conv_cin).
     * Replaces nn.Conv1d / (per output parity) nn.ConvTranspose1d of the temporal U-Nets (reference nn_diffusion/chiunet.py:19-28,
     * jannerunet.py:22-36). */
    int32_t conv_taps, conv_cin, conv_lin, conv_lout, conv_stride, conv_pad;
    /* Split-K for launches that cannot fill the chip with 128 x 128 tiles (few rows, long K): with `partial` != NULL the library
     * may cut K into k_split slices (chosen internally, <= partial_slices), each workgroup writes its raw partial tile to
     * partial[slice][M][N] and a second pass sums the slices in a fixed order and applies the epilogue -- deterministic. */
    float* partial;        /* device scratch of partial_slices * M * N floats, or NULL: never split */
    int32_t partial_slices;
} cdx_gemm_args;
int cdx_gemm_f32(const cdx_gemm_args* args, void* hip_stream);

/* y[m][c] = LN(x[m])[c] [* gamma[c] + beta[c]] [* (1 + scale[m / rows_per_mod][c]) + shift[...]],  C <= 4096.
 * Replaces nn.LayerNorm(+ adaLN `modulate`) (reference dit.py:10-11,33-35,48; idqlmlp.py:14). */
typedef struct cdx_ln_args {
    const float* x;
    float* y;
    const float* gamma; const float* beta;     /* (C) or both NULL */
    const float* scale; const float* shift;    /* (M / rows_per_mod, ldmod) or both NULL */
    int32_t M, C, ldx, ldy, ldmod, rows_per_mod;
    float eps;
    int32_t x_rows;        /* > 0: input row of output row m is m % x_rows (CFG pair sharing one token stream) */
} cdx_ln_args;
int cdx_layernorm_f32(const cdx_ln_args* args, void* hip_stream);

/* GroupNorm over (C/G channels x L positions) per sample on channel-last rows (B*L, C), then activation, FiLM and residual:
 *   y = film_scale * act(gn(x) * gamma + beta) + film_bias + residual
 * film = fa[row_a] + fb[b] (either may be NULL) laid out [scale (C) | bias (C)] (film_mode 1) or [bias (C)] (film_mode 2); row_a is
 * fa_row, or b when fa_per_sample.  Replaces GroupNorm1d + Mish + the FiLM modulation + the block's skip add (reference
 * utils/building_blocks.py:60-76, nn_diffusion/chiunet.py:19-44, jannerunet.py:60-95). */
typedef struct cdx_gn_args {
    const float* x;
    float* y;
    const float *gamma, *beta;       /* (C) */
    const float *fa, *fb;            /* FiLM tables or NULL */
    const float* residual;           /* (B*L, ldr) or NULL */
    int32_t B, L, C, G, ldx, ldy, ldr, ldfa, ldfb, fa_row, fa_per_sample, film_mode, act;
    float eps;
    /* backward only (training): optional (B, C) outputs -- per sample the sums over its positions of dz * x_hat and of dz
     * (dz = d loss / d y * act'): the column sums of these over the batch are d loss / d gamma and d loss / d beta.  Both or neither;
     * needs C / G channels per group to be a power of two <= 256. */
    float *dgamma_part, *dbeta_part;
    /* backward only (ABI 15): optional (C) accumulators -- the kernel ADDS the sums of dz * x_hat / dz to them with float atomics (one
     * per channel and workgroup: a workgroup takes four samples of one group and combines them in LDS first): d loss / d gamma and
     * d loss / d beta themselves, on top of what the buffers held (a parameter's .grad), without the (B, C) staging and its two
     * column-sum launches.  Both or neither, same group-width rule; exclusive with the *_part pair. */
    float *dgamma_sum, *dbeta_sum;
    /* backward only (ABI 17): optional (B, ld_possum) output -- per sample and channel the sum over the sample's positions of
     * d loss / d y: the gradient of a per-sample additive vector behind the activation (the FiLM term of a ResidualBlock, reference
     * jannerunet.py:66), out of the launch that reads d loss / d y anyway.  Same group-width rule. */
    float* dy_possum;
    int32_t ld_possum;
} cdx_gn_args;
int cdx_groupnorm_f32(const cdx_gn_args* args, void* hip_stream);
/* Backward of y = act(gn(x) * gamma + beta) w.r.t. x (classifier guidance, reference classifier/base.py:74-79 asks autograd
 * for d logp / d x; training): same argument block with `x` = the saved forward input, `residual` = d loss / d y (row stride ldr),
 * `y` = d loss / d x; act must be CDX_ACT_MISH or CDX_ACT_NONE; the FiLM fields are ignored. */
int cdx_groupnorm_bwd_f32(const cdx_gn_args* args, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Training step (csrc/cdx_train.hip; SURVEY 8(f4): forward / backward of DiffusionModel.update(), reference diffusionsde.py:94-141).
 * Forward and backward-DATA of the convolutions are cdx_gemm_f32 convolutions (flipped / transposed weights; parity pairs for the strided
 * ones), GroupNorm -> Mish forward / backward-data are the two entries above; these are the sums over the (batch x position) rows.
 * ---------------------------------------------------------------------------------------------- */

/* Weight gradient of a 1-D convolution on channel-last rows:
 *     dw[a][b][t] += sum_{n < batch, m < l_p} p[(n, m)][a] * q[(n, m * stride + t - pad)][b]        (q rows outside [0, l_q): zero)
 * nn.Conv1d(c_in, c_out, k, stride, pad): p = d loss / d y ((batch * l_out, c_out)), q = x ((batch * l_in, c_in)) -> dw (c_out, c_in, k);
 * nn.ConvTranspose1d(c_in, c_out, 4, 2, 1): p = x, q = d loss / d y, stride 2, pad 1 -> dw (c_in, c_out, 4); nn.Linear: taps 1, l = 1.
 * `dw` must be ZEROED by the caller: row slices are combined with float atomics (k_split slices; 0 = chosen by the library). */
typedef struct cdx_wgrad_args {
    const float* p;
    const float* q;
    float* dw;
    int32_t batch, l_p, l_q, ca, cb, taps, stride, pad, ldp, ldq, k_split;
    float* db;             /* optional, zeroed by the caller: db[a] += sum over all rows of p[.][a] -- the bias gradient of an nn.Conv1d
                            * (p = d loss / d y) out of the same launch */
} cdx_wgrad_args;
int cdx_conv_wgrad_f32(const cdx_wgrad_args* args, void* hip_stream);
/* ABI 16: up to CDX_WGRAD_BATCH weight-gradient products in ONE launch (the job table travels as the kernel argument: nothing to
 * upload, capturable as it is).  job[j].k_split >= 1 is the caller's (rows per slice vs float atomics per output element);
 * wg_start[j] = first workgroup of job j, wg_start[n_jobs] = the grid: job j owns ceil(ca / 64) * ceil(cb / 64) * taps * k_split of them.
 * A training step queues the products of all its layers and issues them here (engine/train.py): 65 launches of 4-20 us each -- 38 % of a
 * config-2 update() in round 5 -- become three. */
#define CDX_WGRAD_BATCH 32
typedef struct cdx_wgrad_batch {
    int32_t n_jobs;
    int32_t wg_start[CDX_WGRAD_BATCH + 1];
    cdx_wgrad_args job[CDX_WGRAD_BATCH];
} cdx_wgrad_batch;
int cdx_conv_wgrad_batch_f32(const cdx_wgrad_batch* batch, void* hip_stream);
/* out[c] += sum_r x[r][c] (rows x cols, row stride ld); `out` zeroed by the caller (bias gradients, GroupNorm parameter gradients). */
int cdx_colsum_f32(const float* x, float* out, long long rows, int32_t cols, int32_t ld, void* hip_stream);

/* Backward of cdx_layernorm_f32 (ABI 14; training, csrc/cdx_train.hip): y = xhat * w + b with w = gamma[c] (affine LayerNorm,
 * reference nn_diffusion/idqlmlp.py:14), 1 + scale[m / rows_per_mod][c] (adaLN modulate, dit.py:10-11,33-35,48) or 1 (both NULL).
 *   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w
 * `dyxhat` (M, C; optional): dy * xhat per element -- the caller sums it over rows for the gain gradient (cdx_colsum_f32) or over the
 * rows of a sample for the scale gradient; the shift gradient is the same sum of dy.  C <= 4096. */
typedef struct cdx_ln_bwd_args {
    const float* x;        /* (M, ldx): the forward's input */
    const float* dy;       /* (M, lddy) */
    float* dx;             /* (M, lddx) */
    float* dyxhat;         /* (M, C) or NULL */
    const float* gamma;    /* (C) or NULL */
    const float* scale;    /* (M / rows_per_mod, ldmod) or NULL */
    int32_t M, C, ldx, lddy, lddx, ldmod, rows_per_mod;
    float eps;
} cdx_ln_bwd_args;
int cdx_layernorm_bwd_f32(const cdx_ln_bwd_args* args, void* hip_stream);

/* Backward of cdx_attention_f32 without a mask (ABI 14; training): qkv (B * T, 3 * n_heads * head_dim) as the forward read it,
 * dout (B * T, n_heads * head_dim) -> dqkv (B * T, 3 * n_heads * head_dim); the probabilities are recomputed.  T <= 64, head_dim <= 64
 * (the DiT1d configurations; reference nn_diffusion/dit.py:20,34). */
typedef struct cdx_attn_bwd_args {
    const float* qkv;
    const float* dout;
    float* dqkv;
    int32_t B, T, n_heads, head_dim;
    float scale;
} cdx_attn_bwd_args;
int cdx_attention_bwd_f32(const cdx_attn_bwd_args* args, void* hip_stream);

/* The attention core of nn.MultiheadAttention in TRAINING, forward and backward (ABI 15; csrc/cdx_train.hip): per (sample, head)
 *     P = softmax(q k^T * scale + mask),  out = (P o keep) v
 * with q (B * Tq rows), k / v (B * Tk rows) and out / dout read through their own pointers and row strides (packed qkv rows, a q
 * projection next to a packed kv projection of a memory, ... are the same call); `mask` (Tq, Tk) additive, -inf = not attended, NULL =
 * none; `keep` (B, n_heads, Tq, Tk) the attention-dropout mask as 0 or 1 / (1 - p), drawn by the caller, NULL = no dropout.  What the
 * reference runs through nn.TransformerDecoderLayer / nn.TransformerEncoderLayer in train mode (nn_diffusion/chitransformer.py:108-121,
 * 148-154: causal self-attention, staggered memory cross-attention, attention dropout 0.3) and through nn.MultiheadAttention in
 * dit.py:20,34.  The backward recomputes P from the forward's operands: dq (B * Tq, lddq), dk / dv (B * Tk, lddk / lddv).
 * Tq, Tk <= 64, head_dim <= 64; every row of `mask` must allow at least one key. */
typedef struct cdx_mha_train_args {
    const float* q;
    const float* k;
    const float* v;
    const float* mask;     /* (Tq, Tk) or NULL */
    const float* keep;     /* (B, n_heads, Tq, Tk) or NULL */
    float* out;            /* forward: (B * Tq, ldo) */
    const float* dout;     /* backward: (B * Tq, ldo) */
    float* dq;
    float* dk;
    float* dv;
    int32_t B, Tq, Tk, n_heads, head_dim;
    int32_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float scale;
} cdx_mha_train_args;
int cdx_mha_train_fwd_f32(const cdx_mha_train_args* args, void* hip_stream);
int cdx_mha_train_bwd_f32(const cdx_mha_train_args* args, void* hip_stream);

/* Many small strided re-layouts in ONE launch (ABI 15; training, csrc/cdx_train.hip).  A training step needs every convolution /
 * linear weight in two or three other layouts (taps-major for the implicit GEMM, flipped and transposed for the backward-data
 * convolution, transposed for a Linear's input gradient); as ATen permute / flip / contiguous calls these are ~130 launches of the
 * ~800 of a config-2 step.  Job j copies a 3-D gather into a contiguous block:
 *     dst[(i0 * n1 + i1) * n2 + i2] = src[i0 * s0 + i1 * s1 + i2 * s2]        (strides in elements, negative allowed: src points
 *                                                                                at the element of index (0, 0, 0))
 * `jobs` and `chunks` live on the DEVICE; chunks = n_chunks pairs (job index, chunk index within the job), one workgroup each,
 * CDX_RELAYOUT_CHUNK elements per chunk.  Pure data movement. */
#define CDX_RELAYOUT_CHUNK 2048
typedef struct cdx_relayout_job {
    const float* src;
    float* dst;
    int32_t n0, n1, n2;
    int32_t s0, s1, s2;
} cdx_relayout_job;
int cdx_relayout_f32(const cdx_relayout_job* jobs, const int32_t* chunks, int32_t n_chunks, void* hip_stream);

/* Batch assembly from dataset buffers that live in HBM (third slice: the reference collates a batch on the host --
 * D4RLMuJoCoDataset.__getitem__, cleandiffuser/dataset/d4rl_mujoco_dataset.py:138-151, per item through a torch DataLoader with four
 * worker processes, pipelines/diffuser_d4rl_mujoco.py:34-35,79-83 -- and copies it to the device every step).  Here the padded episode
 * arrays stay on the device and ONE launch gathers every field of a batch:
 *     field f:  out_f[b][s][c] = src_f[row0[b] + s][c]        s < steps_f, c < width_f
 * i.e. a window of steps_f consecutive rows (one contiguous segment of steps_f * width_f floats) per batch item; row0[b] is the item's
 * first row in the flattened (path * max_path_length + position) row space the fields share.  steps 1 = one row per item (the Monte
 * Carlo return of the window's first step; every field of a transition dataset, d4rl_mujoco_dataset.py:214-225).  Pure data movement:
 * the bytes are those of the reference's batch, bit for bit. */
#define CDX_GATHER_MAX_FIELDS 8
typedef struct cdx_gather_field {
    const float* src;      /* device (rows, width), row-major, rows contiguous */
    float* out;            /* device (batch, steps, width) */
    int32_t width, steps;
} cdx_gather_field;
typedef struct cdx_gather_args {
    const int32_t* row0;   /* device [batch] */
    int32_t batch, n_fields;
    long long rows;        /* rows of every src: row0[b] + steps <= rows is the CALLER's contract (checked on the host side of the loader) */
    cdx_gather_field field[CDX_GATHER_MAX_FIELDS];
} cdx_gather_args;
int cdx_gather_windows_f32(const cdx_gather_args* args, void* hip_stream);

/* out[b][t][h*d..] = softmax(q k^T * scale) v per (batch, head); qkv = (B*T, 3*n_heads*head_dim) from in_proj.
 * Replaces the core of nn.MultiheadAttention(batch_first=True) (reference dit.py:20,34).  T <= CDX_ATTN_MAX_T, head_dim <= 64
 * (T <= 64: one wave per (batch, head) on MFMA; longer sequences: streamed keys with an online softmax). */
#define CDX_ATTN_MAX_T 1024
typedef struct cdx_attn_args {
    const float* qkv;
    float* out;            /* (B*T, n_heads*head_dim) */
    int32_t B, T, n_heads, head_dim;
    float scale;
    const float* mask;     /* (T, T) additive mask on the scores, row = query (0 / -inf as nn.Transformer builds them) or NULL */
} cdx_attn_args;
int cdx_attention_f32(const cdx_attn_args* args, void* hip_stream);

/* Cross-attention of T queries against a short memory of S = 1 + n_obs keys per sample: key 0 is a token shared by the whole
 * batch (the timestep token: row `shared_row` of kv_shared, or row b when shared_per_sample), keys 1.. are per-sample rows of
 * kv_rows.  q: (B*T, d); kv_*: (.., 2d) = [k | v] as in_proj[d:3d] produces them; mask: (T, S) additive or NULL.
 * Replaces the memory attention of nn.TransformerDecoderLayer (reference nn_diffusion/chitransformer.py:101-104, 148-155). */
typedef struct cdx_xattn_args {
    const float* q;
    const float* kv_shared;    /* (rows, 2d) */
    const float* kv_rows;      /* (B * n_obs, 2d) */
    const float* mask;
    float* out;                /* (B*T, d) */
    int32_t B, T, n_obs, n_heads, head_dim, shared_row, shared_per_sample;
    float scale;
} cdx_xattn_args;
int cdx_cross_attention_f32(const cdx_xattn_args* args, void* hip_stream);

/* y = act(x) elementwise (batch-invariant embedding vectors: SiLU before adaLN, Mish in map_emb). */
int cdx_act_f32(const float* x, float* y, long long n, int act, void* hip_stream);
/* out = g * act'(pre) elementwise: one factor of an explicit backward pass through an MLP -- `d logp / d x` of the MLP / QGPO energy
 * classifiers without autograd (reference classifier/base.py:74-79 over nn_classifier/mlp.py:10-55); the Linear layers' backward is
 * cdx_gemm_f32 on the transposed weights.  act: none / mish / gelu(erf) / leaky / silu / relu / tanh; `param`: the scale s of a
 * squashed output s * tanh(x / s) (QGPO: s = 10), 1 otherwise. */
int cdx_act_bwd_f32(const float* pre, const float* g, float* out, long long n, int act, float param, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Big-batch sampling loops (csrc/cdx_bigbatch.hip): the whole `sample()` request for the GEMM-shaped denoisers.
 * One call enqueues every kernel of every denoising step on the caller's stream (no host synchronisation, no
 * allocation: the caller owns `workspace`).  The step records are the same cdx_step as above but live in HOST
 * memory here, because the host sequences the launches (all step kinds of cdx_step, EDM kinds 5/6 included).
 * ---------------------------------------------------------------------------------------------- */
typedef struct cdx_sampling {
    int32_t batch;             /* trajectories */
    int32_t hd;                /* floats per trajectory (tokens*in_dim, or the MLP's x dimension) */
    int32_t emb_dim;           /* width of one `temb` row */
    int32_t cond_dim;          /* width of one `cond` row (DiT1d: == emb_dim; residual MLP: obs_dim) */
    const float* temb;         /* device, [max(n_steps,1)][emb_dim] (one row per step record), or [batch][emb_dim] */
    const cdx_step* steps;     /* HOST, [n_steps]; n_steps == 0: one backbone forward x_out <- network(x_in) */
    int32_t n_steps, temb_per_sample, predict_noise;
    int32_t cfg_mode;          /* 0 unconditional, 1 conditional, 2 both: w*c + (1-w)*u on a doubled batch */
    float cfg_w;
    const float* cond;         /* device, [batch][cond_dim] or NULL */
    const float* x_in;         /* (batch, hd) */
    const float* prior;        /* (batch, hd) or NULL */
    const float* fix_mask;     /* (hd) or NULL */
    const float* noise;        /* [n_noise][batch][hd] or NULL */
    const float* x_min;        /* (hd) or NULL */
    const float* x_max;        /* (hd) or NULL */
    float* x_out;              /* (batch, hd) */
    float* workspace;          /* device scratch, >= cdx_*_workspace_floats() floats */
    long long workspace_floats;
    int32_t chunk;             /* trajectories per pass through the loop (0 = whole batch); passes are independent */
} cdx_sampling;

/* DiT1d (reference nn_diffusion/dit.py:14-36 DiTBlock, :39-50 FinalLayer1d, :53-132 DiT1d): all tensors are the
 * checkpoint's own (PyTorch layouts), `pos` is the (tokens, d_model) sinusoidal table of dit.py:122-125. */
typedef struct cdx_dit1d_block {
    const float *ada_w, *ada_b;       /* adaLN_modulation.1: (6d, d), (6d) */
    const float *qkv_w, *qkv_b;       /* attn.in_proj_{weight,bias}: (3d, d), (3d) */
    const float *proj_w, *proj_b;     /* attn.out_proj: (d, d), (d) */
    const float *fc1_w, *fc1_b;       /* mlp.0: (4d, d), (4d) */
    const float *fc2_w, *fc2_b;       /* mlp.3: (d, 4d), (d) */
} cdx_dit1d_block;
/* DiT1Ref (reference dit.py:135-180): before every block the token stream attends to the tokens of a REFERENCE trajectory,
 * x <- cross_attns[i](x, x_ref, x_ref) (nn.MultiheadAttention, no residual); the state rows are [x_ref (in_dim) | x (in_dim)], both
 * halves go through the same x_proj + pos, the output is [x_ref as given | final_layer(x)]. */
typedef struct cdx_dit1ref_cross {
    const float *in_w, *in_b;         /* cross_attns[i].in_proj_{weight,bias}: (3d, d), (3d) -- rows [0,d) q, [d,3d) k | v */
    const float *out_w, *out_b;       /* cross_attns[i].out_proj: (d, d), (d) */
} cdx_dit1ref_cross;
typedef struct cdx_dit1d_weights {
    int32_t tokens, in_dim, emb_dim, d_model, n_heads, depth;
    const float *x_proj_w, *x_proj_b; /* (d, in_dim), (d) */
    const float* pos;                 /* (tokens, d) */
    const float *map0_w, *map0_b;     /* map_emb.0: (d, emb_dim), (d) */
    const float *map2_w, *map2_b;     /* map_emb.2: (d, d), (d) */
    const cdx_dit1d_block* blocks;    /* HOST array [depth] of device pointers */
    const float *fin_ada_w, *fin_ada_b; /* final_layer.adaLN_modulation.1: (2d, d), (2d) */
    const float *fin_w, *fin_b;       /* final_layer.linear: (in_dim, d), (in_dim) */
    const cdx_dit1ref_cross* cross;   /* HOST array [depth] (DiT1Ref: cdx_sampling.hd == tokens * 2 * in_dim) or NULL (DiT1d) */
} cdx_dit1d_weights;
long long cdx_dit1d_workspace_floats(const cdx_dit1d_weights* w, const cdx_sampling* s);
int cdx_dit1d_run(const cdx_dit1d_weights* w, const cdx_sampling* s, void* hip_stream);

/* PearceTransformer (reference nn_diffusion/pearcetransformer.py:8-151; the DBC pipelines' transformer denoiser): S = 2 + To tokens
 * [act_to_input(act_emb(x)) + pos(1) | t_to_input(map_noise(t)) + pos(2) | cond_to_input(condition) + pos(3..)] of width te through
 * n_blocks TransformerEncoderBlocks, then Linear(S * te -> act_dim) on the flattened tokens.  The host folds what is linear:
 *   qkv     = MHA.in_proj o input_to_qkv1                       (te -> 3 td; td = te * n_heads)
 *   o       = bn1a o (attn1_to_fcn o MHA.out_proj) / 1.414       (td -> te), r1 = bn1a scale / 1.414 for the residual path
 *   fc2     = bn1b o attn1_fcn.2 / 1.414                         (4 te -> te), r2 likewise
 * (BatchNorm1d in eval mode is a per-channel affine map; sampling always runs model_ema.eval()), so a block is
 *   a1 = attention(f qkv) o + r1 * f;   f' = gelu(a1 fc1) fc2 + r2 * a1.
 * temb rows are map_noise(t) (emb_dim); cond rows are the flattened (To, emb_dim) condition embedding. */
typedef struct cdx_pearcetf_block {
    const float *qkv_w, *qkv_b;       /* (3 td, te), (3 td) */
    const float *o_w, *o_b, *r1;      /* (te, td), (te), (te) */
    const float *fc1_w, *fc1_b;       /* (4 te, te), (4 te) */
    const float *fc2_w, *fc2_b, *r2;  /* (te, 4 te), (te), (te) */
} cdx_pearcetf_block;
typedef struct cdx_pearcetf_weights {
    int32_t act_dim, To, emb_dim, te, n_heads, n_blocks;
    const float *ae0_w, *ae0_b, *ae2_w, *ae2_b;   /* act_emb.0 / .2: (E, act_dim), (E), (E, E), (E) */
    const float *a2i_w, *a2i_b;       /* act_to_input, pos_embed(1.0) folded into the bias: (te, E), (te) */
    const float *t2i_w, *t2i_b;       /* t_to_input, pos_embed(2.0) folded into the bias */
    const float *c2i_w, *c2i_b;       /* cond_to_input: (te, E), (te) */
    const float* cpos;                /* (To, te): pos_embed(3 .. 3 + To) */
    const cdx_pearcetf_block* blocks; /* HOST array [n_blocks] of device pointers */
    const float *fin_w, *fin_b;       /* final: (act_dim, (2 + To) * te), (act_dim) */
} cdx_pearcetf_weights;
long long cdx_pearcetf_workspace_floats(const cdx_pearcetf_weights* w, const cdx_sampling* s);
int cdx_pearcetf_run(const cdx_pearcetf_weights* w, const cdx_sampling* s, void* hip_stream);

/* ChiTransformer (reference nn_diffusion/chitransformer.py:61-158) with the MLP condition encoder (n_cond_layers == 0):
 * memory = encoder([map_noise(t) | obs_emb(obs)] + cond_pos_emb); decoder layers are nn.TransformerDecoderLayer(norm_first, gelu):
 * h += SA(LN1 h, causal mask); h += CA(LN2 h, memory, memory mask); h += FF(LN3 h); out = head(LN_f h).
 * The memory and its per-layer K/V projections do not depend on x: they are evaluated once per request (observation tokens)
 * and once per step record (timestep token) before the loop.  `temb` rows are map_noise(t) (width d_model); `cond` is
 * (batch, To*obs_dim) or NULL (= zero observations, as the reference substitutes). */
typedef struct cdx_chitf_layer {
    const float *ln1_g, *ln1_b, *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;      /* norm1, self_attn.in_proj (3d,d), out_proj */
    const float *ln2_g, *ln2_b, *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b;      /* norm2, multihead_attn.in_proj (3d,d), out_proj */
    const float *ln3_g, *ln3_b, *ff1_w, *ff1_b, *ff2_w, *ff2_b;                /* norm3, linear1 (4d,d), linear2 (d,4d) */
} cdx_chitf_layer;
/* One layer of the condition encoder when it is an nn.TransformerEncoder (n_cond_layers > 0, reference
 * nn_diffusion/chitransformer.py:91-95; norm_first, GELU, no mask): self-attention over the 1 + To condition tokens, then the FFN. */
typedef struct cdx_chitf_enc_layer {
    const float *ln1_g, *ln1_b, *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;      /* norm1, self_attn.in_proj (3d,d), out_proj */
    const float *ln2_g, *ln2_b, *ff1_w, *ff1_b, *ff2_w, *ff2_b;                /* norm2, linear1 (4d,d), linear2 (d,4d) */
} cdx_chitf_enc_layer;
typedef struct cdx_chitf_weights {
    int32_t Ta, To, act_dim, obs_dim, d_model, n_heads, n_layers;
    const float *act_emb_w, *act_emb_b;      /* (d, act_dim) */
    const float* pos_emb;                    /* (Ta, d) */
    const float *obs_emb_w, *obs_emb_b;      /* (d, obs_dim) */
    const float* cond_pos_emb;               /* (1 + To, d) */
    const float *enc0_w, *enc0_b, *enc2_w, *enc2_b;   /* encoder.0 (4d, d), encoder.2 (d, 4d) -- the MLP encoder (n_enc_layers == 0) */
    const cdx_chitf_layer* layers;           /* HOST array [n_layers] of device pointers */
    const float *lnf_g, *lnf_b, *head_w, *head_b;     /* ln_f, head (act_dim, d) */
    const float* self_mask;                  /* (Ta, Ta) additive */
    const float* memory_mask;                /* (Ta, 1 + To) additive */
    /* Transformer condition encoder: n_enc_layers > 0 replaces the MLP encoder (enc0 / enc2 may then be NULL).  Its self-attention
     * mixes the timestep token with the observation tokens, so the memory is per (sample, step): the encoder and the decoder
     * layers' K/V projections then run once per denoising step instead of once per request. */
    int32_t n_enc_layers;
    const cdx_chitf_enc_layer* enc_layers;   /* HOST array [n_enc_layers] of device pointers, or NULL */
} cdx_chitf_weights;
long long cdx_chitf_workspace_floats(const cdx_chitf_weights* w, const cdx_sampling* s);
int cdx_chitf_run(const cdx_chitf_weights* w, const cdx_sampling* s, void* hip_stream);

/* ChiUNet1d with global conditioning at large batch (reference nn_diffusion/chiunet.py:13-192): every Conv1d is an implicit GEMM
 * over (batch * L) rows, GroupNorm + Mish + FiLM + skip add is one pass (cdx_groupnorm_f32).  Conv weights are PACKED by the
 * host once per weight version: Conv1d (c_out, c_in, k) -> (c_out, k, c_in); ConvTranspose1d(k4 s2 p1) -> two (c_out, 2, c_in)
 * kernels, one per output parity (engine/blocks.py:pack_conv*).  FiLM = cond_encoder.1([mish(time emb) | mish(obs emb)]) is
 * separable, so its time half is evaluated once per step record and its observation half once per request, before the loop.
 * `temb` rows are map_noise(t) (width emb_dim); `cond` is (batch, cond_dim) and required. */
typedef struct cdx_chiunet_block {
    int32_t cin_a, cin_b, cout, groups;      /* input = channel concat [a | b] (cin_b == 0: single input) */
    const float *w1a, *w1b, *b1, *g1, *be1;  /* conv1 split by input part, GroupNorm affine */
    const float *w2, *b2, *g2, *be2;
    const float *film_w, *film_b;            /* cond_encoder.1: (film_out, 2 emb_dim), (film_out) */
    const float *wra, *wrb, *br;             /* residual 1x1 conv split by input part, or all NULL (identity) */
} cdx_chiunet_block;
/* JannerUNet1d(attention=True): LinearAttention (reference nn_diffusion/jannerunet.py:72-95) after the second block of every down
 * level, between the two middle blocks and after the second block of every up level:
 *   xn = LayerNorm_channels(x);  q, k, v = to_qkv(xn) split into heads;  k <- softmax over POSITIONS;  ctx[d][e] = sum_n k[d][n] v[e][n];
 *   out[e][n] = sum_d ctx[d][e] (q[d][n] * dim_head^-0.5);  y = to_out(out) + xn. */
typedef struct cdx_unet_attn {
    const float *ln_g, *ln_b;                /* norm.g / norm.b: (C) */
    const float* qkv_w;                      /* to_qkv 1x1 conv, no bias: (3 * heads * dim_head, C) */
    const float *out_w, *out_b;              /* to_out 1x1 conv: (C, heads * dim_head), (C) */
    int32_t heads, dim_head;
} cdx_unet_attn;
/* the attention core on packed rows: qkv (B * L, 3 * heads * dim_head) = [q | k | v], channel h * dim_head + c -> out (B * L, heads * dim_head);
 * dim_head <= 64, L <= 1024 */
int cdx_linattn_f32(const float* qkv, float* out, int32_t B, int32_t L, int32_t heads, int32_t dim_head, float scale, void* hip_stream);
typedef struct cdx_chiunet_weights {
    int32_t act_dim, Ta, cond_dim, emb_dim, kernel_size, n_levels, cond_predict_scale, model_dim, final_groups;
    /* The same executor serves the unconditional JannerUNet1d (reference nn_diffusion/jannerunet.py:98-201): its blocks add
     * Linear(Mish(emb)) as a per-channel bias (cond_predict_scale = 0), the embedding MLP is emb_dim -> emb_hidden -> emb_out
     * and there is no observation half (cond_dim = 0, film_ld = emb_out).  ChiUNet1d: emb_hidden = 4 emb_dim, emb_out = emb_dim,
     * film_ld = 2 emb_dim. */
    int32_t emb_hidden, emb_out, film_ld;
    const float *map0_w, *map0_b, *map2_w, *map2_b;     /* map_emb.0 (emb_hidden, emb_dim), map_emb.2 (emb_out, emb_hidden) */
    const float *gce_w, *gce_b;                         /* global_cond_encoder (emb_out, cond_dim), or NULL when cond_dim == 0 */
    const cdx_chiunet_block* blocks;                    /* HOST [2 n_levels + 2 + 2 (n_levels - 1)]: downs, mids, ups */
    const float* const* down_w;                         /* HOST [n_levels - 1] packed (C, 3, C) */
    const float* const* down_b;
    const float* const* up_w_even;                      /* HOST [n_levels - 1] packed (C, 2, C) */
    const float* const* up_w_odd;
    const float* const* up_b;
    const float *fin_w, *fin_b, *fin_g, *fin_be;        /* final_conv.0 packed (md, k, md), final_conv.1 GroupNorm */
    const float *out_w, *out_b;                         /* final_conv.3 1x1 (act_dim, md) */
    /* ChiUNet1d with LOCAL conditioning (obs_as_global_cond = False, reference nn_diffusion/chiunet.py:78-82, 153-185): cond_dim = 0,
     * film_ld = emb_out (FiLM from the time embedding alone), the request's cond is (batch, Ta * local_obs_dim) = one observation row
     * per position.  `blocks` then carries two more entries after the main ones -- local_cond_encoder.0 / .1 (local_obs_dim ->
     * model_dim) --; the first one's output is added after the first block of level 0, the second one's, downsampled by lc_down
     * (packed (md, 3, md)), after the first block of the last up level. */
    int32_t local_obs_dim;                              /* 0: no local conditioning */
    const float *lc_down_w, *lc_down_b;
    const cdx_unet_attn* attn;                          /* HOST [2 n_levels]: down levels, middle, up levels -- or NULL (no attention) */
} cdx_chiunet_weights;
long long cdx_chiunet_workspace_floats(const cdx_chiunet_weights* w, const cdx_sampling* s);
int cdx_chiunet_run(const cdx_chiunet_weights* w, const cdx_sampling* s, void* hip_stream);

/* Classifier guidance: (logp, d logp.sum() / d x) of HalfJannerUNet1d by explicit forward + backward launches -- what
 * BaseClassifier.gradients asks torch.autograd for once per denoising step (reference classifier/base.py:74-79,
 * nn_classifier/half_jannerunet.py:102-125, diffusionsde.py:153-173).  Weights are packed by the host (engine/classifier_grad.py):
 * forward convs (c_out, k, c_in); backward-data convs tap-flipped and transposed (c_in, k, c_out); the stride-2 downsample's
 * backward as an even (1 tap) and an odd (2 taps) kernel.  `emb0` = map_noise(t) [+ condition], one row per sample. */
typedef struct cdx_hj_block {
    int32_t cin, cout, k, groups;
    const float *w1, *b1, *w1_bwd, *g1, *be1;
    const float *w2, *b2, *w2_bwd, *g2, *be2;
    const float *emb_w, *emb_b;              /* emb_mlp.1: (cout, model_dim) */
    const float *wr, *br, *wr_bwd;           /* residual 1x1 conv (cout, 1, cin) / (cin, 1, cout), or NULL: identity */
} cdx_hj_block;
typedef struct cdx_hj_down {
    int32_t c;
    const float *w, *b, *bwd_even, *bwd_odd; /* (c, 3, c), (c), (c, 1, c), (c, 2, c) */
} cdx_hj_down;
typedef struct cdx_hjgrad_weights {
    int32_t horizon, in_dim, model_dim, emb_dim, out_dim, fc_hidden, c_last, l_last, n_stages;
    const int32_t* stage_kind;               /* HOST [n_stages]: 0 = next residual block, 1 = next downsample */
    const cdx_hj_block* blocks;              /* HOST */
    const cdx_hj_down* downs;                /* HOST */
    const float *map0_w, *map0_b, *map2_w, *map2_b;
    const float *fc1_wx, *fc1_wx_t, *fc1_we, *fc1_b;   /* final_block.0 split: flattened part in [l][c] order (+ transpose), emb part */
    const float *fc2_w, *fc2_b, *fc2_w_t;              /* final_block.2 (+ transpose) */
} cdx_hjgrad_weights;
long long cdx_hjgrad_workspace_floats(const cdx_hjgrad_weights* w, int32_t batch);
/* emb0_ld: row stride of emb0 in floats; 0 = one row shared by the whole batch (the sampling loop: same t for every sample). */
int cdx_hjgrad_run(const cdx_hjgrad_weights* w, const float* x, const float* emb0, int32_t emb0_ld, int32_t batch, float* logp,
                   float* grad, float* workspace, long long workspace_floats, void* hip_stream);

/* Classifier-guided sampling loop (reference diffusionsde.py:526-594 with w_cg > 0, the configuration every shipped Diffuser
 * pipeline runs): per step record  pred <- backbone(x, t)  [program kernel, forward mode];  (logp, g) <- classifier gradient;
 * pred <- pred - w sigma g (eps prediction) | pred + w sigma^2/alpha g (x0 prediction)  [cg_scale[i], host-frozen];  then clip,
 * solver update and fix-mask exactly as in the unguided loop.  All launches of all steps are enqueued by this one call.
 * Streams: the denoiser launch of each step is issued on a library-owned side stream (one per device, created on first use)
 * forked from and joined back into `hip_stream` with events, so that it overlaps the classifier's launches; every side-stream
 * launch is joined before the call returns, i.e. the caller only ever has to order against `hip_stream`.  Environment
 * CDX_GUIDED_OVERLAP=0 keeps everything on `hip_stream`. */
typedef struct cdx_guided_launch {
    const cdx_unet2_launch* denoiser;    /* a forward-mode launch description of the program kernel (n_steps = 0, its `emb` = the FiLM table
                                          * of ALL steps: one row per step, or batch rows per step with emb_per_traj); emb row / x_in /
                                          * x_out are set per step; NULL when the denoiser runs on the implicit-GEMM executor */
    const cdx_hjgrad_weights* classifier;
    const cdx_step* steps;               /* HOST [n_steps], kinds 0-2 */
    const float* cg_scale;               /* HOST [n_steps]: factor of the gradient added to the prediction at step i */
    int32_t n_steps, batch, hd, predict_noise;
    const float* temb;                   /* device (n_steps, denoiser emb_dim): map_noise(t_i) for the GEMM denoiser; NULL with `denoiser` */
    const float* clf_emb0;               /* device (n_steps, classifier emb_dim): classifier map_noise(t_i) */
    const float *x_in, *prior, *fix_mask, *noise, *x_min, *x_max;
    float* x_out;
    float* workspace;                    /* >= cdx_guided_workspace_floats() */
    long long workspace_floats;
    /* nets whose LDS plan exceeds one workgroup (the shipped antmaze Diffuser: model_dim 64, H = 64): the per-step denoiser forward
     * is the implicit-GEMM U-Net executor (what cdx_chiunet_run runs in forward mode) -- the guided loop is still ONE call */
    const cdx_chiunet_weights* denoiser_gemm;   /* or NULL */
    int32_t denoiser_emb_dim, denoiser_chunk;   /* width of one temb row; trajectories per pass (0 = whole batch) */
} cdx_guided_launch;
long long cdx_guided_workspace_floats(const cdx_guided_launch* g);
int cdx_guided_run(const cdx_guided_launch* g, void* hip_stream);

/* Pre-norm residual MLP = IDQLMlp / NewIDQLMlp (reference nn_diffusion/idqlmlp.py:9-18 ResidualBlock, :21-65, :68-112):
 * features [x | time_mlp(map_noise(t)) | obs] -> affine_in -> n x (h + fc2(mish(fc1(LN(h))))) -> [mish] -> affine_out.
 * `temb` of the request is the table AFTER time_mlp (batch-invariant during sampling). */
typedef struct cdx_resmlp_block {
    const float *ln_g, *ln_b;         /* net.1: (h), (h) */
    const float *fc1_w, *fc1_b;       /* net.2: (4h, h), (4h) */
    const float *fc2_w, *fc2_b;       /* net.4: (h, 4h), (h) */
} cdx_resmlp_block;
typedef struct cdx_resmlp_weights {
    int32_t x_dim, emb_dim, obs_dim, hidden, n_blocks, head_mish;
    const float *in_w, *in_b;         /* affine_in: (h, x_dim+emb_dim+obs_dim), (h) */
    const cdx_resmlp_block* blocks;   /* HOST array [n_blocks] of device pointers */
    const float *out_w, *out_b;       /* affine_out: (x_dim, h), (x_dim) */
} cdx_resmlp_weights;
long long cdx_resmlp_workspace_floats(const cdx_resmlp_weights* w, const cdx_sampling* s);
int cdx_resmlp_run(const cdx_resmlp_weights* w, const cdx_sampling* s, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser side of DiffusionModel.update(): multi-tensor AdamW + EMA + gradient-norm clipping.
 * Replaces, per update() call, torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step() + optimizer.zero_grad() +
 * DiffusionModel.ema_update() (reference diffusion/diffusionsde.py:114-141, diffusion/basic.py:66,83-86): one device table of
 * tensor pointers cut into `chunk_elems`-float chunks, one workgroup per chunk.  AdamW arithmetic = torch/optim/adamw.py's
 * single-tensor path term by term; the host passes step_size = lr / (1 - beta1^t) and bc2_sqrt = sqrt(1 - beta2^t).
 * ---------------------------------------------------------------------------------------------- */
#define CDX_OPT_ADAMW 0   /* g*clip; p *= 1-lr*wd; m,v; p -= step_size*m/(sqrt(v)/bc2_sqrt+eps); [ema = r*ema+(1-r)*p]; [g = 0] */
#define CDX_OPT_EMA 1     /* ema = ema_rate*ema + (1-ema_rate)*p */
#define CDX_OPT_SUMSQ 2   /* norm[0] = sqrt(sum g^2) (fixed-order reduction), norm[1] = min(1, max_norm/(norm[0]+1e-6)) (1 if max_norm <= 0) */
#define CDX_OPT_ZERO 3    /* g = 0 */
#define CDX_OPT_ADAM 4    /* ABI 16: as CDX_OPT_ADAMW with torch.optim.Adam's L2 decay -- g = g*clip + wd*p, no decoupled decay -- the optimiser of
                           * the reference's classifiers (classifier/base.py:24) */
typedef struct cdx_optim_args {
    float* const* p;          /* device [n_tensors]: parameters */
    float* const* g;          /* device [n_tensors]: gradients */
    float* const* m;          /* device [n_tensors]: exp_avg */
    float* const* v;          /* device [n_tensors]: exp_avg_sq */
    float* const* ema;        /* device [n_tensors]: EMA copies (AdamW: NULL = no EMA in this pass) */
    const int64_t* numel;     /* device [n_tensors] */
    const int32_t* chunks;    /* device [n_chunks][2]: (tensor index, chunk index inside the tensor) */
    int32_t n_tensors, n_chunks, chunk_elems, mode;
    float lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, ema_rate;
    float max_norm;           /* > 0: gradients are scaled by norm[1] (written by a preceding CDX_OPT_SUMSQ call) */
    int32_t zero_grad;        /* AdamW: leave the gradients zeroed */
    float* partial;           /* device [n_chunks]: scratch of the norm pass */
    float* norm;              /* device [2] */
} cdx_optim_args;
int cdx_optim_f32(const cdx_optim_args* args, void* hip_stream);

/* Profiling hook: device buffer of [n_workgroups][4] u64 that every following cdx_gemm_f32 launch stamps with s_memtime
 * (start, first tile staged, K loop done, epilogue done); NULL switches it off.  Synchronise before changing it. */
int cdx_gemm_set_trace(unsigned long long* device_buffer);

/* Test hook: runs v_mfma_f32_16x16x4_f32 and v_mfma_f32_4x4x1_16b_f32 on fixed operands
 * (digit-coded lane ids, see csrc/cdx_common.hip) and writes out[4][64][4] so the lane->element maps the kernels
 * rely on are checked on the actual silicon. */
int cdx_probe_mfma_layout(float* out_device, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* CDX_H_ */
