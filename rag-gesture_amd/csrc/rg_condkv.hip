// Condition side of the efficient cross attention in ONE launch per condition (round 6): for every clip b, layer l and head h
//     A[l][b][h] = softmax_tokens(K_h)^T V_h,     [K | V] = LN(xf_b) [W_k | W_v]_l^T + b_l          (32 x 32 per head)
// (mogen/models/attentions/efficient_attention.py:74-90: `key = softmax(key, dim=1)`, `attention = einsum('bnhd,bnhl->bhdl')`;
// raggesture.py:957-1013 builds the condition rows).  Until round 5 this was, per condition, L / 2 GEMMs that wrote K | V of two
// layers as fp32 [rows, 2048] (1.7 GB per batch of 64 clips x (150 + 499 + 1) tokens) and L reductions that read it back
// (kv_reduce_kernel): ~110 launches per set_conditions, 3 ms of kernel time alone and 9-15 ms on the caller's stream of the
// pipeline, where every small launch waits for compute units behind the denoiser's workgroups.  Here a workgroup owns one clip
// (two or more where a clip's tokens need only some of the waves) and the 128 columns [K | V] of TWO heads of one layer: its
// eight waves take 64 token rows each (N <= 512 tokens), the K | V
// tile stays in the accumulators (128 registers), the softmax over the tokens is two exchanges through LDS (column max, column
// sum), P^T V runs on the matrix cores from the accumulators themselves (the C layout of two 16-token blocks IS the operand
// layout of a 32-deep step, for P as A and V as B alike) as bf16 hi + lo products, and the eight partial 32 x 32 matrices are
// added in a fixed order.  Nothing but A (4 KiB per head) is written.
// Operand k-mapping of the projection: a lane reads 32 CONTIGUOUS bytes of a row per pair of k-steps (16 bytes per step), i.e.
// step 2 t takes k = 64 t + 16 g + [0, 8), step 2 t + 1 takes k = 64 t + 16 g + [8, 16) -- any assignment of k to (step, lane
// group) is the same sum as long as A and B agree; this one touches every 128-byte line of X and W in two back-to-back loads.
#include "rg_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int DM = 512, HD = 32, NWV = 8, NTH = NWV * 64;
constexpr int MAX_TOK = NWV * 64;        // tokens per clip and condition
// LDS: column max / sum partials [2][NWV][64] fp32; the workgroup's weight tile as MFMA B fragments [16 k-steps][8 column
// blocks][64 lanes][16 B] = 128 KiB, and -- in the same bytes, once the projection is done -- the waves' partial A matrices
// [NWV][8 blocks][64 lanes] f32x4 (64 KiB)
constexpr int OFF_RED = 0;
constexpr int OFF_W = 2 * NWV * 64 * 4;
constexpr int OFF_PART = OFF_W;
constexpr int LDS_BYTES = OFF_W + 16 * 8 * 64 * 16;
static_assert(NWV * 8 * 64 * 16 <= 16 * 8 * 64 * 16 && LDS_BYTES <= 160 * 1024, "LDS budget");

struct CondKvArgs {
  const unsigned short* xhat;   // bf16 [B][n_tok][512] normalised rows
  const unsigned short* w;      // bf16 [L][1024 (key 512 | value 512)][512], LayerNorm gain folded
  const float* bias;            // fp32 [L][1024], LayerNorm offset folded
  float* out;                   // fp32 A of (layer 0, this condition, first clip): [.. layer_stride ..][B][16][32][32]
  long long layer_stride;       // floats between two layers of `out`
  unsigned short* afrag;        // or null: the same matrices as the sequence-stationary forward's bf16 MFMA A-operand fragments
  long long afrag_layer_stride; // (include/rg_gesture.h rg_seq_args.afrag: [B][8][2 heads][2 column blocks][64 lanes][8] per layer and condition)
  int B, n_tok, L;
};

__device__ __forceinline__ bf16x8 hi_of(const float (&v)[8], float (&rest)[8]) {
  u32x4 h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned short a = __builtin_bit_cast(unsigned short, (__bf16)v[2 * q]), b = __builtin_bit_cast(unsigned short, (__bf16)v[2 * q + 1]);
    h[q] = (unsigned)a | ((unsigned)b << 16);
    rest[2 * q] = v[2 * q] - __uint_as_float((unsigned)a << 16);
    rest[2 * q + 1] = v[2 * q + 1] - __uint_as_float((unsigned)b << 16);
  }
  return __builtin_bit_cast(bf16x8, h);
}
__device__ __forceinline__ bf16x8 bf_of(const float (&v)[8]) {
  u32x4 h;
#pragma unroll
  for (int q = 0; q < 4; ++q) h[q] = rg_pack2_bf16(v[2 * q], v[2 * q + 1]);
  return __builtin_bit_cast(bf16x8, h);
}

__global__ void __launch_bounds__(NTH) cond_kv_kernel(const CondKvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RG_OWN_THE_SIMD();
  float* const sRed = reinterpret_cast<float*>(smem + OFF_RED);
  f32x4* const sPart = reinterpret_cast<f32x4*>(smem + OFF_PART);
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = blockIdx.x >> 3, hp = blockIdx.x & 7;
  const int N = a.n_tok;
  // A clip takes wpc = ceil(N / 64) waves; a workgroup holds cpw = 8 / wpc clips side by side (499 audio tokens: one clip; the
  // 150-token text and speaker conditions: two, six of eight waves busy and the weight tile fetched once for both).  Wave ->
  // (clip slot, 64-row group); a wave without a clip only takes part in the barriers.
  const int wpc = (N + 63) >> 6, cpw = NWV / wpc;
  const int slot = wave / wpc, rgi = wave - slot * wpc;
  const int b = blockIdx.y * cpw + slot;
  const bool active = slot < cpw && b < a.B;
  const unsigned char* X = reinterpret_cast<const unsigned char*>(a.xhat) + (size_t)(active ? b : 0) * N * (DM * 2);
  const unsigned char* W = reinterpret_cast<const unsigned char*>(a.w) + (size_t)l * 1024 * (DM * 2);
  const int r0 = 64 * rgi;

  // accumulators: acc[rb][cb][r] = (K | V)[token r0 + 16 rb + 4 g4 + r][column 16 cb + l15]; columns 0-63 = key columns of heads
  // 2 hp, 2 hp + 1, columns 64-127 = their value columns
  f32x4 acc[4][8];
  {
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      const int col = (cb < 4 ? 64 * hp + 16 * cb : 512 + 64 * hp + 16 * (cb - 4)) + l15;
      const float bv = a.bias[(size_t)l * 1024 + col];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = f32x4{bv, bv, bv, bv};
    }
  }
  // ---- the weight tile into LDS, once, in fragment order (every wave reads all of it, sixteen times over): chunk (s, cb, lane)
  // = 16 bytes of row (column cb's l15-th) at k-step s of the lane group's 32-byte slice
  // the lane's token rows (rows beyond N read row N - 1: finite, masked below): X fragments straight from memory, four k-steps
  // in flight (the first three requested before the weight tile, so that they land behind it)
  unsigned xo[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) xo[rb] = (unsigned)min(r0 + 16 * rb + l15, N - 1) * (DM * 2) + 32 * g4;
  bf16x8 xa[4][4];
  auto load_x = [&](int s, int buf) {     // k-step s: bytes 128 (s >> 1) + 16 (s & 1) of the lane's 32-byte slice
    const unsigned ko = 128 * (s >> 1) + 16 * (s & 1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) xa[buf][rb] = *reinterpret_cast<const bf16x8*>(X + xo[rb] + ko);
  };
  if (active) {
    load_x(0, 0);
    load_x(1, 1);
    load_x(2, 2);
  }
  {
    u32x4 v[16];      // (all sixteen requests of a thread in flight at once: the tile is the first thing a workgroup waits for)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = tid + NTH * q, ln = c & 63, cb = (c >> 6) & 7, st = c >> 9;
      const int col = (cb < 4 ? 64 * hp + 16 * cb : 512 + 64 * hp + 16 * (cb - 4)) + (ln & 15);
      v[q] = *reinterpret_cast<const u32x4*>(W + (size_t)col * (DM * 2) + 128 * (st >> 1) + 32 * (ln >> 4) + 16 * (st & 1));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) *reinterpret_cast<u32x4*>(smem + OFF_W + (tid + NTH * q) * 16) = v[q];
  }
  __syncthreads();
  if (active) {
    // the weight fragments from LDS, each re-read for the next k-step right behind its four products
    const unsigned char* wl = smem + OFF_W + lane * 16;
    bf16x8 wb[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) wb[cb] = *reinterpret_cast<const bf16x8*>(wl + (cb << 10));
#pragma unroll 1
    for (int s4 = 0; s4 < 16; s4 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s4 + u;
        if (s + 3 < 16) load_x(s + 3, (u + 3) & 3);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[u][rb], wb[cb], acc[rb][cb], 0, 0, 0);
          wb[cb] = *reinterpret_cast<const bf16x8*>(wl + ((((s + 1) & 15) * 8 + cb) << 10));      // (behind the last step: step 0 again, unused)
        }
      }
    }
  }
  __syncthreads();      // every wave is done with the weight tile: its bytes become the partial matrices
  // token validity of the lane's rows: bit 4 rb + r
  unsigned vbits = 0;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (active && r0 + 16 * rb + 4 * g4 + r < N) vbits |= 1u << (4 * rb + r);

  // ---- softmax over the tokens, per key column (lane = column 16 cb + l15, cb = 0..3): max, then sum, across the clip's waves
  const int w0 = slot * wpc;      // first wave of this wave's clip
  float cmax[4], cinv[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    float m = -INFINITY;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if ((vbits >> (4 * rb + r)) & 1u) m = fmaxf(m, acc[rb][cb][r]);
    m = rg_xmax4(m);
    if (g4 == 0) sRed[wave * 64 + 16 * cb + l15] = m;
  }
  __syncthreads();
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    float m = -INFINITY;
    if (active)
      for (int w = 0; w < wpc; ++w) m = fmaxf(m, sRed[(w0 + w) * 64 + 16 * cb + l15]);
    cmax[cb] = m;
    const float nm2 = m * -1.44269504088896340736f;
    float s = 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = ((vbits >> (4 * rb + r)) & 1u) ? rg_exp_sub(acc[rb][cb][r], nm2) : 0.f;
        acc[rb][cb][r] = e;
        s += e;
      }
    s = rg_xsum4(s);
    if (g4 == 0) sRed[NWV * 64 + wave * 64 + 16 * cb + l15] = s;
  }
  (void)cmax;
  __syncthreads();
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    float s = 0.f;
    if (active)
      for (int w = 0; w < wpc; ++w) s += sRed[NWV * 64 + (w0 + w) * 64 + 16 * cb + l15];      // fixed order
    cinv[cb] = 1.0f / s;
  }
  (void)cinv;

  // ---- this wave's part of A_h[i][j] = sum_t P[t][i] V[t][j] (h = 2 hp + hh): contraction over its 64 tokens = two 32-deep steps
  // (token blocks 0 | 1, then 2 | 3); P as the A operand and V as the B operand straight from the accumulators, both as bf16
  // hi + lo (three products: hi hi + hi lo + lo hi).  Masked rows: P = 0 (and V finite).
  f32x4 part[2][2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) part[hh][ib][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      bf16x8 ph[2], pl[2], vh[2], vl[2];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        const f32x4 p0 = acc[2 * ks][2 * hh + ib], p1 = acc[2 * ks + 1][2 * hh + ib];
        const float pv[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
        float rest[8];
        ph[ib] = hi_of(pv, rest);
        pl[ib] = bf_of(rest);
        const f32x4 v0 = acc[2 * ks][4 + 2 * hh + ib], v1 = acc[2 * ks + 1][4 + 2 * hh + ib];
        const float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        vh[ib] = hi_of(vv, rest);
        vl[ib] = bf_of(rest);
      }
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          f32x4 d = part[hh][ib][jb];
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[ib], vh[jb], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[ib], vl[jb], d, 0, 0, 0);
          part[hh][ib][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[ib], vh[jb], d, 0, 0, 0);
        }
    }
  }
  // (the MFMA's A operand is indexed [i = l15][k], its B operand [k][j = l15], D[i = 4 g4 + r][j = l15]; P and V sit in the
  //  accumulators as [token][column = l15], i.e. lane l15 holds column i of P resp. column j of V for the lane group's tokens:
  //  exactly A[i = l15][k] and B[k][j = l15] with k = (lane group, register) -- the same enumeration for both)
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) sPart[(wave * 8 + 4 * hh + 2 * ib + jb) * 64 + lane] = part[hh][ib][jb];
  // column sums for the final scaling: A's row i is key column i
  if (active && rgi == 0 && g4 == 0) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) sRed[slot * 64 + 16 * cb + l15] = cinv[cb];
  }
  __syncthreads();
  for (int sl = 0; sl < cpw; ++sl) {
    // wave w adds block w = (hh, ib, jb) of the partial matrices of clip slot sl, in wave order, scales its rows and stores
    const int b = blockIdx.y * cpw + sl;
    if (b >= a.B) break;
    const int hh = wave >> 2, ib = (wave >> 1) & 1, jb = wave & 1;
    f32x4 s = sPart[((sl * wpc) * 8 + wave) * 64 + lane];
    for (int w = 1; w < wpc; ++w) s += sPart[((sl * wpc + w) * 8 + wave) * 64 + lane];
    const int h = 2 * hp + hh;
    float* op = a.out + (size_t)l * a.layer_stride + ((size_t)b * 16 + h) * (HD * HD);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * ib + 4 * g4 + r;
      s[r] *= sRed[sl * 64 + 32 * hh + i];
      op[i * HD + 16 * jb + l15] = s[r];
    }
    if (a.afrag) {
      // fragment (head h = 2 hp + hh, column block jb): lane (jj = l15, g = g4) holds A[i][16 jb + jj] for i = 4 g + e (e < 4: this
      // workgroup's ib = 0 block) and 16 + 4 g + e - 4 (ib = 1): each of the two waves stores its four values (8 bytes)
      unsigned short* fp = a.afrag + (size_t)l * a.afrag_layer_stride + (((((size_t)b * 8 + hp) * 2 + hh) * 2 + jb) * 64 + lane) * 8 + 4 * ib;
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      *reinterpret_cast<u32x2*>(fp) = u32x2{rg_pack2_bf16(s[0], s[1]), rg_pack2_bf16(s[2], s[3])};
    }
  }
}

}  // namespace

extern "C" int rg_cond_kv(rg_handle* h, const void* xhat_bf16, const void* w_bf16, const float* bias, float* out, long long layer_stride,
                          void* afrag_bf16, long long afrag_layer_stride, int B, int n_tok, int L, void* stream) {
  RG_REQUIRE(h, xhat_bf16 && w_bf16 && bias && out, "null pointer");
  RG_REQUIRE(h, B >= 1 && L >= 1 && n_tok >= 1 && n_tok <= MAX_TOK, "unsupported shape (1 <= tokens per clip <= 512)");
  static rg_attr_once lds_once;
  if (!rg_reserve_lds(lds_once, cond_kv_kernel, LDS_BYTES)) {
    h->err = "rg_cond_kv: cannot reserve LDS";
    return RG_ERR_HIP;
  }
  CondKvArgs a;
  a.xhat = reinterpret_cast<const unsigned short*>(xhat_bf16);
  a.w = reinterpret_cast<const unsigned short*>(w_bf16);
  a.bias = bias; a.out = out; a.layer_stride = layer_stride;
  a.afrag = reinterpret_cast<unsigned short*>(afrag_bf16); a.afrag_layer_stride = afrag_layer_stride;
  a.B = B; a.n_tok = n_tok; a.L = L;
  const int wpc = (n_tok + 63) / 64, cpw = NWV / wpc;      // waves per clip, clips per workgroup
  hipLaunchKernelGGL(cond_kv_kernel, dim3(8 * L, (B + cpw - 1) / cpw), dim3(NTH), LDS_BYTES, rg_stream(stream), a);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}
