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

int rg_version(void);
int rg_create(rg_handle** out, int device);
void rg_destroy(rg_handle* h);
const char* rg_last_error(rg_handle* h);

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

/* Exemplar splice: copy token rows [r0,r1) of src[b_src] into rows [q0,q1) of dst[b_dst]
 * for the upper block and the hands block (row offset n_lat+1)
 *   (diffusion_architecture.py:386-407).  src/dst are [*,T,D]. */
int rg_splice_rows(rg_handle* h, const float* src, float* dst, int T, int D, int n_lat,
                   int b_src, int b_dst, int r0, int r1, int q0, int q1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RG_GESTURE_H */
