/*
 * rg_gesture.h -- C ABI of the MI355X (gfx950) HIP extension for the RAG-Gesture
 * inference hot path.
 *
 * The reference (m-hamza-mughal/RAG-Gesture) is 100% Python on PyTorch eager ops; it has
 * no FFI of its own.  Each entry point below therefore cites the reference Python
 * function whose arithmetic it replaces (path:line under the reference tree), and
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name
 *     ends in _host.  Buffers are owned by the caller (torch tensors in the host code).
 *   - every launch goes to `stream` (a hipStream_t passed as void*); no entry point
 *     synchronises, allocates or reads back, so all of them are hipGraph-capturable.
 *   - return 0 on success, negative rg_status otherwise; rg_last_error() gives the text.
 *   - fp32 activations are row-major; "token rows" M = batch_rows * T (T = 43 latent tokens).
 */
#ifndef RG_GESTURE_H
#define RG_GESTURE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_handle rg_handle;

enum rg_status {
  RG_OK = 0,
  RG_ERR_INVALID = -1,   /* bad argument (null pointer, unsupported shape) */
  RG_ERR_HIP = -2,       /* a HIP runtime call or launch failed */
  RG_ERR_NO_DEVICE = -3
};

/* ABI version of THIS header: bumped whenever an argument block (rg_seq_args, ...) or an entry point changes shape.  A caller
 * compares it with rg_version() of the library it loaded (rag-gesture_amd/capi.py load_library does and refuses a mismatch).
 * 110 (round 6): rg_seq_args gains `form`; rg_seqx_forward, rg_lane_form.  100 -> 105 were the unversioned states of rounds
 * 1-5 (round 5 changed rg_seq_args -- gbuf in, l0 / l1 out -- and the afrag layout without a bump: ADVICE r05). */
#define RG_VERSION 111
int rg_version(void);
int rg_create(rg_handle** out, int device);
void rg_destroy(rg_handle* h);
const char* rg_last_error(rg_handle* h);
/* Compute units of the handle's device (tile-shape policies: one workgroup per CU while a launch fits). */
int rg_num_cus(rg_handle* h);

/* ---------------------------------------------------------------- sampler (elementwise)
 * DDIM update with x0-prediction, eta = 0:
 *   eps = (c_recip * x - x0) / c_recipm1 ;  x_out = x0 * c_a + c_b * eps
 * Forward step: c_a = sqrt(abar_prev), c_b = sqrt(1 - abar_prev)
 *   (gaussian_diffusion.py:693-697 `_predict_eps_from_xstart`, :981-1001 `ddim_sample`).
 * Inversion step: c_a = sqrt(abar_next), c_b = sqrt(1 - abar_next)
 *   (gaussian_diffusion.py:1003-1040 `ddim_reverse_sample`).
 * Coefficients are the reference's fp64 tables cast to fp32 by the host.
 * x_out may alias x.  n = number of floats. */
int rg_ddim_update(rg_handle* h, const float* x, const float* x0, float* x_out, int64_t n,
                   float c_recip, float c_recipm1, float c_a, float c_b, void* stream);

/* Classifier-free mix of the denoiser head output followed by the DDIM update, fused:
 *   x0[b,t,:] = js[t]*w_c*out[b,t,:] + (1/js[t])*w_u*out[B+b,t,:]
 *   (raggesture.py:1086-1111 `forward_test` CFG mix; w_c = both+text, w_u = retr+none)
 * then rg_ddim_update on x0.  out is [2B,T,D] (cond rows first), x/x_out [B,T,D],
 * js = per_joint_scale mask [T].  x0_out may be NULL. */
int rg_cfg_ddim_update(rg_handle* h, const float* out, const float* x, float* x_out, float* x0_out,
                       const float* js, int B, int T, int D, float w_c, float w_u,
                       float c_recip, float c_recipm1, float c_a, float c_b, void* stream);
/* The same on a sub-range of a batch: out_cond / out_uncond point at the conditional and the classifier-free rows of the
 * same B clips (they are not B*T*D apart when the denoiser batch holds more sequences); x_out2 (may be NULL) receives a
 * second copy of the updated latent (the inversion stores every level and keeps the working rows in place). */
int rg_cfg_ddim_update_rows(rg_handle* h, const float* out_cond, const float* out_uncond, const float* x, float* x_out,
                            float* x_out2, const float* js, int B, int T, int D, float w_c, float w_u,
                            float c_recip, float c_recipm1, float c_a, float c_b, void* stream);

/* The same CFG mix followed by one ancestral (DDPM) step, `inference_type="ddpm"`:
 *   x_out = (c1 * x0 + c2 * x) + sigma * noise
 *   (gaussian_diffusion.py:479-501 `q_posterior_mean_variance` with model_mean_type START_X, :560-570 fixed_large
 *   variance, :795-803 `p_sample`: sigma = exp(0.5 * log_variance), 0 at the last step).  x_out may alias x. */
int rg_cfg_ddpm_update(rg_handle* h, const float* out, const float* x, const float* noise, float* x_out,
                       const float* js, int B, int T, int D, float w_c, float w_u, float c1, float c2, float sigma,
                       void* stream);

/* Everything between two denoiser forwards of a chain that advances a sampling loop (group a: n_a clips) and an inversion
 * loop (group b: n_b clips) in the same launches, in ONE launch: group a: rg_cfg_ddim_update_rows of this step, then -- for the
 * NEXT step, on the rows in_seq_next marks -- rg_guidance_update (g_iter_next iterations, numel = n_a * T * D) and
 * rg_inseq_replace; group b: rg_cfg_ddim_update_rows with its own coefficients and a second copy of the result.  The same
 * arithmetic as those entry points, operation for operation (gaussian_diffusion.py:910-1001 `ddim_sample`, :1003-1040
 * `ddim_reverse_sample`, :1263-1273, :1351-1378 guidance). */
typedef struct rg_glue_args {
  const float* out_c_a;    /* denoiser output, conditional rows of group a [n_a][T][D] */
  const float* out_u_a;    /* ... classifier-free rows of group a */
  float* x_a;              /* latent of group a, updated in place */
  const float* out_c_b;
  const float* out_u_b;
  float* x_b;              /* latent of group b, updated in place */
  float* x_b_copy;         /* or NULL: second copy of group b's result (the inversion level kept) */
  const float* in_seq_next;   /* or NULL: the next step's in_seq of group a [n_a][T][D] */
  const float* noise_next;    /* its randn_like(in_seq) draw */
  const float* js;         /* per_joint_scale [T] */
  int n_a, n_b, T, D;
  int g_iter_next;         /* insertion-guidance iterations of the next step (0: none) */
  float wc_a, wu_a, c_recip_a, c_recipm1_a, ca_a, cb_a;   /* CFG weights and DDIM coefficients of group a's step */
  float wc_b, wu_b, c_recip_b, c_recipm1_b, ca_b, cb_b;   /* ... of group b's */
  float lr, s_ab_next, s_1mab_next;
} rg_glue_args;
int rg_cobatch_glue(rg_handle* h, const rg_glue_args* args_host, void* stream);

/* In-sequence replacement (outpainting / exemplar insertion / prev-latent chaining):
 *   m[r] = any(in_seq[r,:] != 0);  x[r,:] = m ? s_ab*in_seq[r,:] + s_1mab*noise[r,:] : x[r,:]
 *   (gaussian_diffusion.py:934-947 in `ddim_sample`, :459-477 `q_sample`).  rows = B*T. */
int rg_inseq_replace(rg_handle* h, float* x, const float* in_seq, const float* noise,
                     int rows, int dim, float s_ab, float s_1mab, void* stream);

/* Insertion-guidance update: g_iter gradient steps on mse(x*m, in_seq) w.r.t. x,
 *   x <- x - lr * (2/numel) * ((x*m - in_seq) * m),  m as above, numel = rows*dim
 *   (gaussian_diffusion.py:1263-1273 `retrieval_guidance`, :1351-1378 the autograd loop). */
int rg_guidance_update(rg_handle* h, float* x, const float* in_seq, int rows, int dim,
                       int g_iter, float lr, void* stream);

/* All exemplar splices of a batch in one launch: for entry i, rows [r0, r0 + nrows) of exemplar e (upper block and hands
 * block, row offset n_lat + 1) of inv [S][Ep][T][D] go into rows [q0, q0 + nrows) of clip b -- level `lvl` into start_noise
 * [B][T][D] (rg_splice_rows) and, when invl [S][B][T][D] is given, every level into it (rg_splice_rows_rep)
 * (diffusion_architecture.py:386-407).  The entries' destinations do not overlap (raggesture.py:703-731 de-overlaps them). */
#define RG_SPLICE_MAX 64
typedef struct rg_splice_table {
  int n;
  int e[RG_SPLICE_MAX], b[RG_SPLICE_MAX], r0[RG_SPLICE_MAX], q0[RG_SPLICE_MAX], nrows[RG_SPLICE_MAX];
} rg_splice_table;
int rg_splice_many(rg_handle* h, const rg_splice_table* tab_host, const float* inv, float* start_noise, float* invl,
                   int T, int D, int n_lat, int lvl, int S, int Ep, int B, void* stream);

/* Exemplar splice: copy token rows [r0,r1) of src[b_src] into rows [q0,q1) of dst[b_dst]
 * for the upper block and the hands block (row offset n_lat+1)
 *   (diffusion_architecture.py:386-407).  src/dst are [*,T,D]. */
int rg_splice_rows(rg_handle* h, const float* src, float* dst, int T, int D, int n_lat,
                   int b_src, int b_dst, int r0, int r1, int q0, int q1, void* stream);
/* The same splice repeated for nrep (diffusion level) slabs: slab s reads batch item
 * b_src + s*src_rep_stride and writes b_dst + s*dst_rep_stride (the 50 per-timestep inverted
 * latents, diffusion_architecture.py:399-407). */
int rg_splice_rows_rep(rg_handle* h, const float* src, float* dst, int T, int D, int n_lat,
                       int b_src, int b_dst, int r0, int r1, int q0, int q1, int nrep,
                       int src_rep_stride, int dst_rep_stride, void* stream);

/* ---------------------------------------------------------------- fused bf16-MFMA GEMM
 * out[M,N] = epilogue( A'[M,K] * W[N,K]^T ), bf16 operands, fp32 accumulate (MFMA 16x16x32).
 * It replaces every nn.Linear on the hot path together with the elementwise ops around it:
 *   A' can be built on the fly from fp32 sources: identity cast, LayerNorm
 *   (efficient_attention.py:29,31,36,72-74 `self.norm(x)`), or the StylizationBlock front half
 *   LN -> *(1+scale)+shift -> SiLU (stylization_block.py:36-39);
 *   the epilogue adds bias / a token-periodic table (positional embeddings,
 *   diffusion_transformer.py:646-659) / a residual (efficient_attention.py:44,100;
 *   diffusion_transformer.py:86), applies GELU (diffusion_transformer.py:85) or the per-head
 *   softmax over head_dim=32 of the queries (efficient_attention.py:32,78), and can emit
 *   per-row partial (sum, sumsq) so the consumer's LayerNorm needs no extra pass.
 * The descriptor is a HOST struct read at call time. */

#define RG_MAX_SEG 4

// A-operand transforms applied while an fp32 source tile is staged into LDS as bf16
enum { RG_A_IDENT = 0, RG_A_LN = 1, RG_A_STYL = 2 };

typedef struct rg_a_segment {
  const float* src;    // fp32 source, row-major
  int ld;              // row stride of src (floats)
  int mode;            // RG_A_IDENT / RG_A_LN / RG_A_STYL
  const float* stats;  // [rows][nparts][2] partial (sum, sumsq) over the segment's seg_len columns
  int nparts;
  int pad_;
  const float* gamma;  // [seg_len] LayerNorm weight (LN, STYL)
  const float* beta;   // [seg_len] LayerNorm bias
  const float* scale_shift;  // STYL: [2*seg_len] = AdaLN scale | shift for this (step, layer, block)
} rg_a_segment;

typedef struct rg_gemm_desc {
  int M, N, K;
  int a_is_bf16;          // 1: A is bf16 [M, lda] (nseg = 0; or nseg = 1 with seg[0].mode = RG_A_STYL: the rows are the
                          //    bf16 copy of a block output y and the operand is SiLU((y - mean) * rstd * gain + offset):
                          //    the StylizationBlock front half with gain = gamma (1 + scale), offset = beta (1 + scale)
                          //    + shift folded by the caller into seg[0].gamma / .beta (16-byte aligned, K floats each;
                          //    .scale_shift NULL, .src unused) and mean / rstd from seg[0].stats; applied once per
                          //    element in LDS by the LDS-DMA kernel: K <= 512, K % 64 == 0); 0: fp32 segments
  const void* A;          // bf16 A (a_is_bf16)
  int lda;
  int a_row_mod;          // >0: A row index = row % a_row_mod (row-duplicating GEMMs)
  int seg_len;            // K columns per fp32 segment (K = nseg * seg_len, last may be short)
  int nseg;
  rg_a_segment seg[RG_MAX_SEG];
  int gb_group;           // >0: column tiles [g*gb_group, (g+1)*gb_group) form group g: gamma/beta of every fp32
  int gb_stride;          //     segment, or the bf16 A pointer, are offset by g * gb_stride elements (several
                          //     projections of differently normalised copies of the same rows in one launch)
  const void* W;          // bf16 [Np, ldw]  (rows = output features, K contiguous, zero padded)
  int ldw;
  int act;                // 0 none, 1 GELU(erf), 2 ReLU
  const float* bias;      // [N] or null
  const float* tbias;     // [tb_period, N] or null: + tbias[(row % tb_period) * N + col]
  int tb_period;
  int softmax_cols;       // columns [0, softmax_cols) get a softmax over each group of 32 columns
  const float* residual;  // fp32 [M, ldr] or null
  int ldr;
  int out_bf16;           // 1: out is bf16, 0: fp32
  void* out;
  int ldo;
  int ldo2;               // row stride (elements) of out2
  float* stats_out;       // [M][N/64][2] partial (sum, sumsq) of the final fp32 output, or null
  const void* W_lo;       // null, or bf16 [Np, ldw] = bf16(W - float(bf16(W))): selects the precise
                          // "bf16x3" mode (hi*hi + hi*lo + lo*hi), fp32 A segments only
  void* out2;             // null, or bf16 [M, ldo2]: a second, bf16-rounded copy of the output (the next
                          // GEMM's A operand, written by the producer instead of converted by every consumer tile)
  const float* ln_stats;  // null, or [M][ln_nparts][2] partial (sum, sumsq) of the fp32 rows whose bf16 copy is A:
  const float* ln_c1;     //   LayerNorm folded into the epilogue.  With W' = W diag(gamma) packed as the weight,
  int ln_nparts;          //   c1[n] = sum_k W'[n][k] and bias[n] = b[n] + sum_k W[n][k] beta[k]:
  int split_col;          //   out = rstd * (A W'^T - mean * c1) + bias  ==  LN(x) W^T + b   (mean/rstd over K columns)
                          // split_col > 0 (multiple of 128): output columns >= split_col go ONLY to out2 (bf16, column
                          // index - split_col), columns below it only to out: a GEMM with an fp32 and a bf16 consumer
  int tile_n;             // 0 / 128: default tile width; 64: 64x64 tiles (bf16 A, aligned shapes only): more, smaller
  int pad4_;              //   workgroups for single-round GEMMs; stats_out then holds N/64 partials per row
} rg_gemm_desc;

int rg_gemm(rg_handle* h, const rg_gemm_desc* desc_host, void* stream);
/* n GEMMs that share a shape signature (M, N, K, operand kinds, epilogue features) in one launch: descriptor i belongs to
 * the workgroups with blockIdx.y = i (the four body-part VAEs run the same layer on different weights: 4x fewer dependent
 * launches in the front end, which is what it costs beside running denoiser chains).  descs_host: array of n descriptors;
 * groups that do not share a signature, or n > 4, are launched one by one.  Same result bits as n rg_gemm calls. */
int rg_gemm_grouped(rg_handle* h, const rg_gemm_desc* descs_host, int n, void* stream);

/* The A-operand prologue of rg_gemm as a standalone pass: out[row, s*seg_len + k] =
 * bf16(f_s(src_s[row,k])) for nseg fp32 segments (identity / LayerNorm / stylization front half,
 * same rg_a_segment descriptors, HOST array).  Used in front of the K = 4*512 ca_mix GEMM so the
 * SiLU prologue runs once per element instead of once per column tile.
 * Rows [m_cond, M) are the classifier-free rows: for the first unc_nseg segments they copy
 * unc_tab[flag][s*seg_len + k] (bf16; flag = 1 where qmask[s][row] == 0) instead of reading the source:
 * with cond_type 0 the cross-attention output is the value bias for every token
 * (efficient_attention.py:83-90), so its stylized form depends on the weights and the
 * timestep only and is tabulated at load time.  m_cond = M disables this. */
int rg_stylize(rg_handle* h, const rg_a_segment* segs_host, int nseg, int seg_len, int M, void* out_bf16, int ldo,
               int m_cond, int unc_nseg, const void* unc_tab_bf16, const float* qmask, void* stream);
/* rg_stylize for a batch whose sequences sit at TWO diffusion steps (the sampling rows of one batch of clips and the
 * inversion rows of the next batch's exemplars in the same launches): rows are [2 CFG halves][nseq sequences][T tokens];
 * sequences >= split of either half take their (scale | shift) from scale_shift_b_host[s] (HOST array of nseg device
 * pointers) instead of segs[s].scale_shift.  split >= nseq: one group, as rg_stylize. */
int rg_stylize_groups(rg_handle* h, const rg_a_segment* segs_host, int nseg, int seg_len, int M, void* out_bf16, int ldo,
                      int m_cond, int unc_nseg, const void* unc_tab_bf16, const float* qmask,
                      const float* const* scale_shift_b_host, int T, int nseq, int split, void* stream);

/* Kernel selection hook for tests / tuning: 0 = auto, 1 = generic register-staged kernel only (any
 * shape), 2 = prefer the LDS-DMA ring kernel, 3 = prefer the depth-4 register-staged kernel, 4 = prefer the
 * 128-row big-tile kernel (bf16 A; 128 x 256 tiles), 5 = never use it, 6 = prefer it with 128 x 128 tiles,
 * 7 = 128 x 128 tiles on a ring of 2 at two workgroups per CU. */
int rg_set_gemm_path(rg_handle* h, int path);
/* Tuning knob of this handle: waves per workgroup of the LDS-DMA GEMM kernel (0 = auto by shape, 4 or 8; default 0). */
int rg_set_gemm_waves(rg_handle* h, int waves);

/* Measurement aid (bench.py roofline): between begin and end every rg_gemm launch is bracketed by
 * HIP events on its stream; end synchronises and returns launch count, summed kernel time and summed
 * algorithmic FLOPs (2*M*N*K) for one kernel variant (0: fp32-source A, 1: bf16 A, 2: bf16x3; 3: rg_seq_forward
 * launches, FLOPs = the unit GEMMs and attention products of the T token rows of every sequence). */
int rg_profile_begin(rg_handle* h);
int rg_profile_end(rg_handle* h, int variant, int64_t* launches_host, double* total_ms_host,
                   double* total_flops_host);

/* out[i,:] = table[idx[i],:]  (nn.Embedding lookup of speaker ids, diffusion_transformer.py:544-548).
 * idx is int64 on device; dim % 4 == 0. */
int rg_gather_rows(rg_handle* h, const float* table, const int64_t* idx, float* out, int n, int dim,
                   void* stream);

/* ---------------------------------------------------------------- linear attention
 * Self-attention core of EfficientSelfAttention (efficient_attention.py:32-41) for R batch rows
 * of T tokens: qkv is [R*T, ldqkv] fp32 with q (already softmaxed over head_dim by the GEMM
 * epilogue) in columns [0,D), k in [D,2D), v in [2D,3D); src_mask [R,T] (0 = masked token:
 * `key + (1-mask)*-1e6` and `value*mask`).  Writes y [R*T, ldy] fp32 and per-row partial
 * LayerNorm statistics stats[R*T][D/128][2] (sum, sumsq over each 128-column head group).
 * perm (device, nperm ints, or NULL): launch order -> work item (row * D/128 + head group, -1 = idle
 * block); lets the caller place a row group on the XCD whose L2 already holds its rows.
 * mode 1: P^T V and Q A on the matrix cores (bf16 hi + lo operand pairs, fp32 accumulate: the bf16
 * production path); 0: exact fp32 VALU products (precision = "fp32"); 2: as 1 and y leaves as bf16 [R*T, ldy]
 * (T <= 48): the A operand of the SA-out GEMM, which applies the stylization in LDS (rg_gemm_desc.seg). */
int rg_sa_attention(rg_handle* h, const float* qkv, int ldqkv, const float* src_mask, void* y, int ldy,
                    float* stats, int R, int T, int D, const int* perm, int nperm, int mode, void* stream);

/* Cross-attention core of EfficientCrossAttention for ncond parallel conditions
 * (efficient_attention.py:90-98; diffusion_transformer.py:105-118): y3[:, c*D:(c+1)*D] = Q_c A_c
 * with Q_c = q3[:, c*D:(c+1)*D] (softmaxed), A_c = Apre[c][row] ([H][32][32], from rg_kv_reduce) for
 * rows < Rc and Aunc[c] for the classifier-free rows [Rc,R) (cond_type 0: V = value(0) = bias, so
 * A[d][l] = b_v[l]; efficient_attention.py:83-90).
 * qmask [ncond][R][T] or NULL: where 0, y is rounded exactly as the reference's fp32
 * `y + (1-query_mask)*-1e6` rounds it (grid 1/16) so that the following LayerNorm sees the
 * same values up to the shift.  stats layout [ncond][R*T][D/128][2].
 * perm as for rg_sa_attention; work item = (row * ncond + cond) * D/128 + head group. */
int rg_ca_attention(rg_handle* h, const float* q3, const float* Apre, const float* Aunc, const float* qmask,
                    float* y3, float* stats, int R, int Rc, int T, int D, int ncond, const int* perm, int nperm,
                    void* stream);

/* rg_ca_attention + the stylization front half (StylizationBlock: LN, *(1+scale)+shift, SiLU;
 * stylization_block.py:30-47) of the three cross-attention outputs in one launch, one workgroup per
 * (conditional row group, condition): out[b*T+n][c*D + col] (bf16, row stride ldo) for the Rc conditional
 * row groups (y = Q A on the matrix cores, Q and A as bf16 hi + lo pairs: At_bf16 = rg_split_transpose_bf16 of
 * the rg_kv_reduce output, [ncond][Rc][H][2][32][32]); the Ru classifier-free row groups [Rc, Rc+Ru) receive unc_tab[flag][0 .. ncond*D) (flag = 1
 * where qmask == 0), see rg_stylize.  q3 holds the conditional rows only ([Rc*T][ncond*D]);
 * qmask is [ncond][Rc+Ru][T] or NULL; gamma/beta [ncond][D]; scale_shift [ncond][2*D]. */
int rg_ca_stylize(rg_handle* h, const float* q3, const void* At_bf16, const float* qmask, const float* gamma,
                  const float* beta, const float* scale_shift, const void* unc_tab_bf16, void* out_bf16, int ldo,
                  int Rc, int Ru, int T, int D, int ncond, void* stream);
/* The same for two groups of sequences at different diffusion steps: row groups >= split of the conditional rows take
 * scale_shift_b, those >= split of the classifier-free rows unc_tab_b (see rg_stylize_groups). */
int rg_ca_stylize_groups(rg_handle* h, const float* q3, const void* At_bf16, const float* qmask, const float* gamma,
                         const float* beta, const float* scale_shift, const void* unc_tab_bf16, void* out_bf16, int ldo,
                         int Rc, int Ru, int T, int D, int ncond, const float* scale_shift_b, const void* unc_tab_b_bf16,
                         int split, void* stream);

/* At[m] = (bf16(A[m]^T), bf16(A[m]^T - hi)) for n_mat fp32 32x32 matrices: the B-operand layout rg_ca_stylize reads. */
int rg_split_transpose_bf16(rg_handle* h, const float* A, void* At_bf16, int n_mat, void* stream);

/* A[b][h] = softmax_over_tokens(K[b,:,h,:])^T V[b,:,h,:]  (efficient_attention.py:82-90) for B rows
 * of N conditioning tokens; kv is [B*N, ldkv] fp32 with k in columns [0,D) and v in [D,2D).
 * A is [B][H][32][32].  Loop invariant over the 50 DDIM steps: run once per clip. */
int rg_kv_reduce(rg_handle* h, const float* kv, int ldkv, float* A, int B, int N, int D, void* stream);

/* Partial LayerNorm statistics of fp32 rows that no GEMM produced (the gathered speaker
 * embeddings): stats[row][dim/64][2] = (sum, sumsq) over each 64-column part. */
int rg_row_stats(rg_handle* h, const float* x, float* stats, int rows, int dim, void* stream);

/* Guard of the folded LayerNorm (rg_gemm_desc.ln_stats): *max_ratio = max(*max_ratio, max over rows of mean^2 / var)
 * from the partial (sum, sumsq) statistics [rows][nparts][2] of K-wide rows.  *max_ratio must hold a non-negative
 * float (zero it first).  The folded form multiplies bf16(x) with x un-normalised: its error grows with |mean| / std
 * (efficient_attention.py:29,72-74 normalise first); the host reads the figure once per session and uses the
 * LayerNorm pre-pass instead when it is large. */
int rg_ln_guard(rg_handle* h, const float* stats, int rows, int nparts, int K, float* max_ratio, void* stream);

/* Exact fp32 linear for load-time tables: out[M,N] = f_out( f_in(a)[M,K] w[N,K]^T + bias ),
 * f = SiLU if the flag is set (time_embed MLP, diffusion_transformer.py:404-408, and the 40
 * StylizationBlock emb_layers, stylization_block.py:35, which depend only on the timestep). */
int rg_linear_f32(rg_handle* h, const float* a, const float* w, const float* bias, float* out, int M,
                  int N, int K, int silu_in, int silu_out, void* stream);

/* Condition side of the efficient cross attention, one launch per condition (csrc/rg_condkv.hip): for every clip b, layer l and
 * head h,  A[l][b][h] = softmax_tokens(K_h)^T V_h  with  [K | V] = xhat_b W_l^T + bias_l
 * (mogen/models/attentions/efficient_attention.py:74-90; xhat = the condition rows after the LayerNorm WITHOUT its affine,
 * which is folded into W / bias: LN_l(x) W_l^T + b_l = xhat (W_l diag(gamma_l))^T + (b_l + W_l beta_l)).
 * xhat bf16 [B][n_tok][512], w bf16 [L][1024 = key 512 | value 512][512], bias fp32 [L][1024]; out fp32: the A matrices of
 * (layer 0, clip 0), [16 heads][32][32] per clip, layer l at out + l * layer_stride.  n_tok <= 512, 16 heads of 32.
 * afrag_bf16 (or NULL): the same matrices once more as rg_seq_args.afrag of (layer 0, this condition, clip 0) -- bf16 MFMA
 * A-operand fragments [8][2 heads][2 column blocks][64 lanes][8] per clip, layer l at + l * afrag_layer_stride elements.
 * Replaces rg_gemm (K | V of two layers as fp32 [rows][2048]) + rg_kv_reduce per layer: nothing but A is written. */
int rg_cond_kv(rg_handle* h, const void* xhat_bf16, const void* w_bf16, const float* bias, float* out, long long layer_stride,
               void* afrag_bf16, long long afrag_layer_stride, int B, int n_tok, int L, void* stream);

/* ---------------------------------------------------------------- one denoiser forward, sequence-stationary
 * ReGestureTransformer.forward at inference (raggesture.py:1041-1085 `forward_test` up to the CFG mix,
 * diffusion_transformer.py:620-668, :105-127 `DecoderLayer`, :74-87 `FFN`, efficient_attention.py:23-45, 62-102,
 * stylization_block.py:29-40) in ONE launch of 2B workgroups: workgroup s owns sequence s (T <= 48 token rows; the B
 * conditional sequences first, then the B classifier-free ones) from the joint embedding to the output head.  The fp32
 * residual stream stays in registers, the bf16 MFMA operand panels in LDS; the only memory traffic is the weight stream,
 * which every wave fetches for itself by LDS-DMA in the order it consumes it (csrc/rg_seq.hip).  D = 512, 16 heads x 32,
 * FF = 1024, L <= 8, bf16 operands / fp32 accumulate, LayerNorm statistics exact in fp32.
 * Streams (DEVICE, built once per model / per clip batch by the host, rag-gesture_amd/seqfwd.py):
 *   wstream  bf16  [NU][8 waves][64][1 KiB]   NU = 16 L + 2 unit GEMMs of 512 x 512 (embed; per layer k, v, q, sa_out,
 *            mix_x, (q3_c, mix_c) x 3, ff1_0, ff2_0, ff1_1, ff2_1, ffn_out; head).  Fragment (wave w, step s, block j) is
 *            the MFMA operand image of W[64 w + 16 j + (lane & 15)][32 s + 8 (lane >> 4) + e], e = 0..7.  LayerNorm gains
 *            are folded into the k / v / q / q3 weights (W diag(gamma)), their betas into the biases.
 *   pstream  fp32  [S][NU][8][4][64]          per step and unit, for the wave's 64 features: vector 0 = bias, 1 / 2 =
 *            stylization gain gamma (1 + scale) / offset beta (1 + scale) + shift of the block the unit feeds
 *            (mix_x: vector 1 = row sums of the bf16 weight, for the un-normalised x segment).
 *   ustream  fp32  [S][L][8][2 KiB]           classifier-free sequences: sum_k W_mix_c[n][k] * unc_tab[flag][c][k] for
 *            (c, flag) = (0,0) (0,1) (1,0) (1,1) | (2,0) (2,1): their cross-attention output is a constant of (step, layer).
 *   afrag    bf16  [L][3][B][8][2 heads][2 column blocks][64 lanes][8]   A = softmax_N(K)^T V of every clip,
 *            condition and head (rg_kv_reduce) as MFMA A-operand fragments: element e of lane (jj, g) is
 *            A[i][16 jb + jj], i = e < 4 ? 4 g + e : 16 + 4 g + e - 4.
 * Clips >= split run at step index step_b (two diffusion loops sharing the launch: sampler.cobatched_loop).
 * dump / dump_stage: diagnostics (stage 1: after the embedding; 2 / 3 / 4: after the self-attention / cross-attention /
 * FFN block of layer dump_layer; 10: self-attention output y; 11-13: cross-attention y of condition 0-2), the T-layout
 * registers go to dump [2B][48][512] and the workgroup exits. */
typedef struct rg_seq_args {
  const void* wstream;
  const void* pstream;
  const void* ustream;
  const void* afrag;
  const float* x;          /* fp32 [B][T][512] latent at this step */
  const float* tbias;      /* fp32 [T][512] positional tables */
  const float* src_mask;   /* fp32 [2B][T] */
  const float* qmask;      /* fp32 [3][2B][T] */
  float* head;             /* fp32 [2B][T][512] result */
  float* dump;
  float* xbuf;             /* rg_seq2_forward: fp32 [workgroups][2][8][12][64][4] round trips of the two sequences' tiles; rg_seq_forward: unused */
  void* gbuf;              /* rg_seq2_forward: bf16 [workgroups][4][2][48 KiB] panel images; rg_seq_forward: unused */
  const int* form;         /* rg_seqx_forward: DEVICE flag of the launching lane (rg_lane_form): 0 = two sequences per workgroup
                              as rg_seq2_forward, 1 = one per workgroup as rg_seq_forward; the other two entry points ignore it */
  int L, B, T, S;          /* layers, clips, tokens, steps in pstream / ustream */
  int step, step_b, split; /* clips [0, split) at step, clips [split, B) at step_b */
  int dump_stage, dump_layer;
  int pairs;               /* 0: one workgroup per sequence (2 B workgroups); 1: one workgroup per clip runs the conditional
                              sequence, then its classifier-free twin (B workgroups, 1.7x as long: less CU time per forward,
                              for callers that run several narrow launches side by side) -- same results bit for bit */
  int* glue_ctr;           /* or NULL.  DEVICE int [B], zero-initialised once by the caller and then left to the launches: with
                              it, the forward ends with what rg_cobatch_glue does between two forwards of a loop (`glue` below;
                              n_a + n_b == B, n_b may be 0; its out_* pointers are rows of `head`, its x_* rows of `x`) -- the
                              workgroup that finishes the second of a clip's two sequences updates that clip's rows, so a loop
                              step is ONE launch (csrc/rg_tail.h).  Same arithmetic, operation for operation. */
  rg_glue_args glue;
} rg_seq_args;

int rg_seq_forward(rg_handle* h, const rg_seq_args* args_host, void* stream);
int rg_seq2_forward(rg_handle* h, const rg_seq_args* args_host, void* stream);

/* The same forward with the launch form chosen ON THE DEVICE when the launch starts (csrc/rg_seqx.hip): launched with the
 * grid of the one-sequence form, it runs rg_seq2_forward's workgroups (the surplus ones leave at once) or rg_seq_forward's,
 * as args->form says -- same bits either way.  rg_lane_form is the arbitration in front of it, on the same stream: state is
 * int [nlanes][RG_LANE_STRIDE] in device memory (one 128-byte line per lane, written by that lane's launches only),
 * zero-initialised by the caller and shared by the lanes (streams) of one pipeline; state[l][0] = workgroups lane l's current
 * launches hold, state[l][1] = lane l's flag (args->form = &state[l][1]).  The lane takes the wide form
 * (wide_wgs workgroups) when the other lanes' workgroups + wide_wgs <= budget (the chip's compute units), else the narrow one
 * (narrow_wgs); narrow_wgs == wide_wgs publishes a fixed-form launch's load, 0 / 0 an idle lane (the end of a chain).
 * Nothing in the reference corresponds to this: its launches are whatever PyTorch eager issues
 * (mogen/models/architectures/diffusion_architecture.py:283-452 drives one batch at a time). */
#define RG_LANE_STRIDE 32
int rg_seqx_forward(rg_handle* h, const rg_seq_args* args_host, void* stream);
int rg_lane_form(rg_handle* h, int* state, int lane, int nlanes, int narrow_wgs, int wide_wgs, int budget, void* stream);

/* Sequence-stationary body-part VAE encoder: the whole skip-transformer stack of `TransformerVAE.encode_to_dist`
 * (mogen/models/transformers/gesture_vae.py:111-193: chunk sequences of frame_chunk_size frames + the two distribution
 * tokens; mogen/models/utils/detr_utils.py:101-152 SkipTransformerEncoder, :335-393 TransformerEncoderLayer.forward_post with
 * torch.nn.MultiheadAttention) in ONE launch, one workgroup per two chunk sequences; replaces the per-op launch chain
 * (rg_gemm / rg_mha_bf16 / rg_layernorm) for the shape it is specialised for: latent_dim 512, 4 heads, ff_size 1024, GELU,
 * post-norm, S <= 24 tokens, bf16 operands with fp32 accumulation, fp32 residual stream / LayerNorm / softmax.
 *   x       fp32 [nseq][S][512]   the embedded sequences (skel_embedding + the distribution tokens + positional table)
 *   out     fp32 [nseq][S][512]   encoder output behind the final LayerNorm (rows 0 / 1 = mu / logvar: rg_vae_reparam)
 *   wstream bf16 [NU][8 waves][64 fragments][64 lanes][8]   unit GEMMs (512 x 512) in consumption order, per block:
 *           [skip W[:, :D], skip W[:, D:]] (output blocks only), Q (x 1/sqrt(128)), K, V, out_proj, linear1[:512],
 *           linear2[:, :512], linear1[512:], linear2[:, 512:];  NU = 8 (2 nb + 1) + 2 nb
 *   pstream fp32 [NU + 1][8 waves][4][64]   per unit: vector 0 = bias (zero for the second half of a split unit), vectors
 *           1 / 2 = gamma / beta of the LayerNorm that follows the unit (out_proj: norm1, linear2 first half: norm2);
 *           slot NU: vectors 0 / 1 = gamma / beta of the encoder's final norm
 *   xbuf    fp32 [ceil(nseq / 2)][nb][8][12][64][4]   the skip stack (states behind the input blocks)
 *   nb      input blocks = output blocks ((num_layers - 1) / 2 for odd num_layers, num_layers / 2 for even: detr_utils.py:108)
 *   dump / dump_block: diagnostics (state behind block dump_block as fp32 [workgroup][48][512]); dump_block < 0: off. */
typedef struct rg_venc_args {
  const void* wstream;
  const void* pstream;
  const float* x;
  float* out;
  float* xbuf;
  float* dump;
  int nseq, S, nb, dump_block;
} rg_venc_args;

int rg_venc_forward(rg_handle* h, const rg_venc_args* args_host, void* stream);
/* n <= 4 independent stacks (the four body-part VAEs: different weights, inputs and widths of nothing -- the stacks share D = 512)
 * in ONE launch: args_host[n], gridDim.y = n. */
int rg_venc_forward_grouped(rg_handle* h, const rg_venc_args* args_host, int n, void* stream);

/* Block-fused body-part VAE decoder of the "all_encoder" architecture (mogen/models/transformers/gesture_vae.py:195-239
 * `decode`: the skip-transformer ENCODER stack over cat(z [10], zero queries [150]) = 160 tokens with pos = query_pos, i.e.
 * q = k = x + pos, v = x (detr_utils.py:335-393), num_heads * 8 = 32 heads of 16): one launch per block instead of nine,
 * a sequence cut into four tiles of 40 rows (one workgroup each), the tiles' keys / values exchanged through L2 between the
 * launches.  Launch `step` (0 .. 2 nb + 1) runs [attention, out_proj + norm1, FFN + norm2 of block step - 1] -> [skip push or
 * skip linear of block step] -> [Q, K, V of block step]; launch 0 only projects, the last one ends with the stack's final norm.
 * Same weight / parameter streams as rg_venc_forward (Q pre-scaled by 1/sqrt(16)); specialised for latent_dim 512, 4 x 8
 * heads, ff_size 1024, GELU, post-norm, 160 tokens, bf16 operands with fp32 accumulation.
 *   x    fp32 [nseq][160][512]   residual stream: the stack's input before step 0, its output (behind the final norm) after the
 *                                last step, updated in place by every launch
 *   pos  fp32 [nseq][160][512]   query_pos (constant over the blocks)
 *   qimg bf16 [4 nseq][48 KiB]   a tile's Q operand panel as it lies in LDS (handed from launch to launch)
 *   kbuf bf16 [2][nseq][160][512]   keys, row-major;   vt bf16 [2][nseq][512][160]   values, feature-major; both double-buffered
 *                                by launch parity (a launch's faster tiles write block `step`'s while slower ones still read
 *                                block step - 1's)
 *   xbuf fp32 [4 nseq][nb][8][12][64][4]   the skip stack;   dump: diagnostics (fp32 [4 nseq][48][512]: the state a launch starts
 *                                its projection part from) or NULL */
typedef struct rg_vdec_args {
  const void* wstream;
  const void* pstream;
  float* x;
  const float* pos;
  void* qimg;
  void* kbuf;
  void* vt;
  float* xbuf;
  float* dump;
  int nseq, nb, step, pad_;
} rg_vdec_args;

int rg_vdec_step(rg_handle* h, const rg_vdec_args* args_host, void* stream);
int rg_vdec_step_grouped(rg_handle* h, const rg_vdec_args* args_host, int n, void* stream);

/* ---------------------------------------------------------------- body-part VAEs + rotations
 * Softmax multi-head attention core of torch.nn.MultiheadAttention for short sequences
 * (detr_utils.py:364-366, 427-433): o[b,i,h,:] = softmax_j(q[b,i,h,:].k[b,j,h,:]/sqrt(hd)) v[b,j,h,:].
 * q [B*Sq, ldq], k/v [B*Sk, ld], head h in columns [h*hd, (h+1)*hd); Sk <= 192; no padding mask
 * (the reference always passes full-length clips, gesture_vae.py:124-160). */
int rg_mha(rg_handle* h, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
           int ldo, int B, int H, int Sq, int Sk, int hd, void* stream);
/* Same operation on the matrix cores for the bf16 path (K, V rounded to bf16, Q and the softmax
 * probabilities as bf16 hi + lo pairs, fp32 accumulation): hd in {16, 32, 64, 128}, Sk <= 192 (Sk <= 512 at hd = 64: wav2vec2 windows), row
 * strides multiples of 4 floats.  out_is_bf16 = 1: o is bf16 [B*Sq, ldo] (the A operand of the out-projection). */
int rg_mha_bf16(rg_handle* h, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, void* o, int ldo,
                int out_is_bf16, int B, int H, int Sq, int Sk, int hd, void* stream);

/* nn.LayerNorm(dim), eps 1e-5, fp32 rows (detr_utils.py norm1/norm2/norm3, encoder/decoder norm). */
int rg_layernorm(rg_handle* h, const float* x, const float* gamma, const float* beta, float* out, int rows, int dim,
                 void* out_bf16, void* stream);

/* LayerNorm(x + residual) with the caller's eps (residual may be NULL): the post-norm steps of the conditioning encoders
 * (BERT eps 1e-12, wav2vec2 1e-5; transformers BertSelfOutput / Wav2Vec2EncoderLayer).  out_bf16: optional bf16 copy. */
int rg_layernorm_res(rg_handle* h, const float* x, const float* residual, const float* gamma, const float* beta, float* out,
                     int rows, int dim, float eps, void* out_bf16, void* stream);

/* ---------------------------------------------------------------- conditioning feature extraction
 * tools/longform_synthesis.py:64-94 computes, per 10-second window, BERT-base-cased hidden states (sum of the last four
 * layers) of the window's transcript and the wav2vec2-base-960h last hidden state [499, 768] of its audio.  Both encoders
 * run on rg_gemm / rg_mha_bf16 / rg_layernorm_res (host: rag-gesture_amd/features.py); these are the remaining pieces:
 *   rg_embed_sum3: out[t] = word[ids[t]] + type0 + pos[t]  (BertEmbeddings before its LayerNorm; ids int64 on device)
 *   rg_time_groupnorm_gelu: x [T][C] fp32 -> GELU(GroupNorm_{groups = C}(x)) as bf16 and / or fp32 (either output may
 *     be NULL; normalisation over TIME per channel: layer 0 of the wav2vec2 feature extractor, feat_extract_norm =
 *     "group"); workspace: 2 * C floats
 *   rg_im2col_grouped: out[g][t][k * C/groups + ci] = x[t + k - pad][g * C/groups + ci] (zero outside [0, T)), bf16: the
 *     patch matrices of the positional convolution (Conv1d(768, 768, 128, padding 64, groups 16)), one GEMM per group. */
int rg_embed_sum3(rg_handle* h, const int64_t* ids, const float* word, const float* pos, const float* type0, float* out,
                  int L, int dim, void* stream);
int rg_time_groupnorm_gelu(rg_handle* h, const float* x, const float* gamma, const float* beta, void* out_bf16,
                           float* out_f32, int T, int C, float eps, float* workspace, void* stream);
int rg_im2col_grouped(rg_handle* h, const float* x, void* out_bf16, int T, int C, int groups, int ksize, int pad, void* stream);

/* Grouped forms: n <= 4 problems of ONE shape in one launch (host arrays of n device pointers; everything else shared).
 * The four body-part VAEs run the same layer sequence on different weights and rows: issuing a layer of all four as one
 * launch divides the number of dependent launches of a VAE encode / decode by four (vae.py records the parts' launch
 * sequences and zips them).  Result bits equal the single calls. */
int rg_layernorm_grouped(rg_handle* h, int n, const float* const* x, const float* const* gamma, const float* const* beta,
                         float* const* out, int rows, int dim, void* const* out_bf16, void* stream);
int rg_add_rows_grouped(rg_handle* h, int n, const float* const* a, const float* const* b, float* const* out, int64_t n_elems,
                        int64_t period, void* stream);
int rg_copy_rows_grouped(rg_handle* h, int n, const float* const* src, float* const* dst, int groups, int nrows_per, int dim,
                         int rows_src_per, int src_row0, int rows_dst_per, int dst_row0, void* stream);
int rg_mha_bf16_grouped(rg_handle* h, int n, const float* const* q, int ldq, const float* const* k, int ldk,
                        const float* const* v, int ldv, void* const* o, int ldo, int out_is_bf16, int B, int H, int Sq, int Sk,
                        int hd, void* stream);
/* out[i] = a[i] + b[i % period]: positional embeddings / `with_pos_embed` (detr_utils.py:357-358). */
int rg_add_rows(rg_handle* h, const float* a, const float* b, float* out, int64_t n, int64_t period, void* stream);

/* Row gather/scatter between [groups, rows_per, dim] tensors (token prepend, latent slicing:
 * gesture_vae.py:150-158, 214-216; diffusion_transformer.py:239-254, 273-278). rows_src_per = 0
 * broadcasts the same source rows to every group. */
int rg_copy_rows(rg_handle* h, const float* src, float* dst, int groups, int nrows_per, int dim, int rows_src_per,
                 int src_row0, int rows_dst_per, int dst_row0, void* stream);

/* Column copy dst[r, dcol+c] = src[r, scol+c]; columns flagged in rel_mask are made relative to the
 * first frame of their clip (frames rows per clip): `trans[:,:,0] -= trans[:,0:1,0]`
 * (diffusion_transformer.py:231-232). */
int rg_copy_cols(rg_handle* h, const float* src, int ld_src, int scol, float* dst, int ld_dst, int dcol, int rows,
                 int ncols, int frames, unsigned rel_mask, void* stream);

/* Reparameterisation with explicit noise, scattered into the [B,T,D] diffusion latent:
 * latent[b, row_off + c, :] = mu + exp(logvar)^0.5 * eps, mu/logvar = tokens 0/1 of the encoder
 * output enc [B*n_chunks, seq, D] (gesture_vae.py:173-193). */
int rg_vae_reparam(rg_handle* h, const float* enc, int seq, const float* eps, float* latent, int B, int n_chunks,
                   int D, int T, int row_off, void* stream);

/* axis-angle [rows, joints*3] -> 6D [rows, joints*6] written at column col_off of out, and back
 * (rotation_conversions.py:416-430 + 535-550; 511-532 + 433-447). */
int rg_aa_to_6d(rg_handle* h, const float* aa, int ld_in, float* out, int ld_out, int col_off, int rows, int joints,
                void* stream);
int rg_6d_to_aa(rg_handle* h, const float* d6, int ld_in, int col_off, float* out, int ld_out, int rows, int joints,
                void* stream);

/* ---------------------------------------------------------------- caller-side packing
 * tools/visualize.py:208-213: pred_motion[..., part_mask] = pred_part for upper / lower / hands / face in one
 * pass.  The masks are whole joints: src_part[j] in {0 upper, 1 lower, 2 hands, 3 face, -1 none (zeros)},
 * src_joint[j] = joint index inside that part.  out is [rows, joints*3]. */
int rg_scatter_joints(rg_handle* h, const float* upper, int ld_u, const float* lower, int ld_l, const float* hands,
                      int ld_h, const float* face, int ld_f, const int* src_part, const int* src_joint, float* out,
                      int rows, int joints, void* stream);
/* tools/visualize.py:266-291 / longform_synthesis.py:714-741: axis-angle -> 6D, F.interpolate(mode='linear',
 * scale_factor=scale, align_corners=False) along time, 6D -> axis-angle, fused.  aa [B,n,joints*3] ->
 * out [B,n*scale,joints*3].  rg_interp_linear: the same interpolation for plain features (expressions, trans). */
int rg_interp_aa(rg_handle* h, const float* aa, float* out, int B, int n, int joints, int scale, void* stream);
int rg_interp_linear(rg_handle* h, const float* x, float* out, int B, int n, int dim, int scale, void* stream);
/* tools/longform_synthesis.py:431-476: blend of a new window with the last `overlap` frames of the motion so
 * far.  Every frame of the window goes axis-angle -> 6D -> axis-angle; on frames t < overlap the 6D value is
 * prev6D*(1-w_t) + new6D*w_t, w = torch.linspace(0, 1, overlap).  prev_tail [B,overlap,joints*3], cur/out
 * [B,n,joints*3].  rg_blend_linear blends plain features in place on cur[:, :overlap]. */
int rg_blend_aa(rg_handle* h, const float* prev_tail, const float* cur, float* out, int B, int n, int joints, int overlap,
                void* stream);
int rg_blend_linear(rg_handle* h, const float* prev_tail, float* cur, int B, int n, int dim, int overlap, void* stream);

/* ---------------------------------------------------------------- retrieval sweep
 * Score of every DB entry for ONE query relation (sense, connective), float64, the reference's
 * operation order (rag/discourse_retrieval.py:86-222): +2 sense present, +4 exact connective among the
 * entry's relations of that sense, +3 same speaker, + mean over those relations (prominence known on
 * both sides) of 4/(1+2|p_db - p_q|).  The DB is integer-coded CSR: rel_off[n_entries+1] indexes
 * rel_sense / rel_conn / rel_prom (NaN = unknown); q_conn = -1 if the connective is not in the DB
 * vocabulary; q_prom NaN = unknown.  top_out[e] = index (within the entry) of the relation whose
 * bounds are reported (`top_rel_idx`), -1 if the sense is absent. */
int rg_discourse_scores(rg_handle* h, const int* spk, const int* rel_off, const int* rel_sense,
                        const int* rel_conn, const double* rel_prom, int n_entries, int q_sense, int q_conn,
                        int q_spk, double q_prom, double* score_out, int* top_out, void* stream);

/* rg_discourse_scores for all queries of a batch in ONE launch: params[q] = (sense code, connective code, speaker id,
 * query prominence or NaN) as four doubles (the integer codes are exact); score_out / top_out are
 * [n_queries][n_entries].  Replaces the per-query loop over raggesture.py:498-511 -> discourse_retrieval. */
int rg_discourse_scores_batched(rg_handle* h, const int* spk, const int* rel_off, const int* rel_sense,
                                const int* rel_conn, const double* rel_prom, int n_entries, const double* params,
                                int n_queries, double* score_out, int* top_out, void* stream);

/* Score of every DB entry for ONE query gesture label (type, word) of the gesture_type retrieval method
 * (rag/gesture_type_retrieval.py:41-117) and of the llm method (rag/llm_retrieval.py:278-419), float64: over the
 * entry's non-beat labels of the query type (CSR lab_off / lab_type / lab_word, words integer coded): +2 type present,
 * +spk_bonus same speaker (2 for gesture_type, 1 for llm), +5 exact word or +3/(1+2*max_r word_sim[lab_word[r]]);
 * word_sim[v] = get_word_similarity_score(DB word v, query word) computed by the caller; q_word = -1 if the query
 * word is not a DB word.  llm only: lab_prom (NaN = unknown; NULL for gesture_type) and q_prom (NaN = unknown) add
 * the mean of 4/(1+2|lab_prom - q_prom|) and move the reported label to the one of smallest difference.
 * sim_f32 != 0: the similarity model returns numpy float32 and NumPy >= 2 promotion applies, which makes the
 * reference's score arithmetic float32 from the similarity term on; 0: float64 (python floats, or NumPy < 1.24).
 * top_out = index (within the entry's non-beat labels) of the label whose bounds are reported, -1 if the type is
 * absent. */
int rg_gesture_scores(rg_handle* h, const int* spk, const int* lab_off, const int* lab_type, const int* lab_word,
                      const double* lab_prom, const double* word_sim, int n_entries, int q_type, int q_word, int q_spk,
                      double spk_bonus, double q_prom, int sim_f32, double* score_out, int* top_out, void* stream);

/* Word similarity of the gesture_type / llm methods as the reference effectively computes it (rag/utils.py:239-272:
 * the word2vec / fasttext models are undefined, so every call returns `fuzz.partial_ratio(db_word, query_word) / 100`,
 * fuzzywuzzy 0.18, pure-python SequenceMatcher flavour): out[v] = partial_ratio(vocab word v, query) / 100 for the
 * n_words strings vocab[v][0 .. vocab_len[v]) (int32 code points, row stride `stride`) -- the word_sim vector
 * rg_gesture_scores takes.  query_host: HOST int32 code points.  Strings longer than rg_partial_ratio_max_len() code
 * points give NaN (the host computes those with the standard library's difflib). */
int rg_partial_ratio(rg_handle* h, const int* vocab, const int* vocab_len, int n_words, int stride,
                     const int* query_host, int query_len, double* out, void* stream);
int rg_partial_ratio_max_len(void);

/* Candidate selection for the ranking walk of rag/discourse_retrieval.py:224-300: the reference sorts
 * all scores and visits entries until it holds 10, so only entries with score >= the 10th largest
 * score (with multiplicity; all ties included) and score > 0 can be visited.  Appends those
 * entries (index, top_rel_idx, score) in arbitrary order to out_*[0..cap) and leaves their count in
 * *cursor (a count > cap means the lists were truncated).  workspace: rg_select_workspace_doubles(n)
 * doubles.  Comparisons only: exact. */
int rg_select_top_scores(rg_handle* h, const double* score, const int* top, int n_entries, double* workspace,
                         int* cursor, int cap, int* out_idx, int* out_top, double* out_score, void* stream);
int rg_select_workspace_doubles(int n_entries);
/* The same for n_queries score rows at once (three launches in all): score / top [n_queries][n_entries], workspace
 * [n_queries][rg_select_workspace_doubles(n)], cursor [n_queries], out_* [n_queries][cap]. */
int rg_select_top_scores_batched(rg_handle* h, const double* score, const int* top, int n_entries, int n_queries,
                                 double* workspace, int* cursor, int cap, int* out_idx, int* out_top,
                                 double* out_score, void* stream);
/* rg_discourse_scores_batched + rg_select_top_scores_batched in TWO launches without the [n_queries][n_entries] score / relation
 * arrays (rag/discourse_retrieval.py:86-300 for all query relations of a batch): the scores stay in registers, the per-slice
 * top-10 lists give every query's threshold, and a second pass over the integer-coded DB (L2-resident) appends the entries
 * that reach it.  Same arithmetic and comparisons as the two-step form: identical survivor sets.  params [n_queries][4]
 * doubles as for rg_discourse_scores_batched; workspace [n_queries][rg_select_workspace_doubles(n)]; cursor [n_queries]
 * (zeroed by the call); out_* [n_queries][cap]. */
int rg_discourse_select_fused(rg_handle* h, const int* spk, const int* rel_off, const int* rel_sense, const int* rel_conn,
                              const double* rel_prom, int n_entries, const double* params, int n_queries, double* workspace,
                              int* cursor, int cap, int* out_idx, int* out_top, double* out_score, void* stream);

/* Tie-break similarity (rag/utils.py:109-121): out[j] = mean_{i < min(Lq, L_c)} q[i,:].feats_c[i,:]
 * for candidate entries c = cand[j]; feats is the ragged [sum L, dim] fp32 token-feature table with
 * int64 row offsets feat_off[n_entries+1]; fp64 accumulation. */
int rg_text_diag_sim(rg_handle* h, const float* q, int Lq, const float* feats, const int64_t* feat_off,
                     const int* cand, int n_cand, int dim, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RG_GESTURE_H */
