// Retrieval sweep kernels (replicated exemplar DB resident in HBM).
//
//  rg_discourse_scores : the per-DB-entry categorical + prominence score of
//      rag/discourse_retrieval.py:86-222 for one query relation, in float64 with the reference's
//      operation order (sense +2, connective +4, speaker +3, mean of 4/(1+2|dp|) over the entry's
//      relations of that sense), plus the index of the relation whose bounds are returned
//      (`top_rel_idx`).  One thread per DB entry over integer-coded CSR metadata: pure HBM
//      streaming (<= ~40 B per entry).
//  rg_text_diag_sim    : the tie-break similarity of rag/utils.py:86-132,
//      mean(diag(q[Lq,768] . db[Ld,768]^T)) = mean over i < min(Lq,Ld) of q[i].db[i], for a list
//      of candidate entries; fp32 products accumulated in fp64, one workgroup per candidate, 16 B
//      per lane coalesced reads of the candidate's token features (the reference computes the full
//      Lq x Ld matrix and takes its diagonal).
#include "rg_common.h"

namespace {

__device__ __forceinline__ void discourse_score_value(
    const int e, const int* __restrict__ spk, const int* __restrict__ rel_off, const int* __restrict__ rel_sense,
    const int* __restrict__ rel_conn, const double* __restrict__ rel_prom, int q_sense, int q_conn,
    int q_spk, double q_prom, double& score_ret, int& top_ret) {
  const int r0 = rel_off[e], r1 = rel_off[e + 1];
  double score = 0.0;
  int top = -1;
  int first = -1, conn_hit = -1;
  for (int r = r0; r < r1; ++r) {
    if (rel_sense[r] == q_sense) {
      if (first < 0) first = r;
      if (conn_hit < 0 && q_conn >= 0 && rel_conn[r] == q_conn) conn_hit = r;
    }
  }
  if (first >= 0) {
    score += 2.0;
    top = first;
    bool chosen = false;
    if (conn_hit >= 0) {
      score += 4.0;
      top = conn_hit;
      chosen = true;
    }
    if (spk[e] == q_spk) score += 3.0;
    double sum = 0.0, best = 0.0;
    int cnt = 0, best_r = -1;
    if (q_prom == q_prom) {  // query prominence known (not NaN)
      for (int r = first; r < r1; ++r) {
        if (rel_sense[r] != q_sense) continue;
        const double p = rel_prom[r];
        if (p != p) continue;
        const double diff = fabs(p - q_prom);
        sum += 4.0 / (1.0 + 2.0 * diff);
        if (best_r < 0 || diff < best) {  // first minimum = stable sort of the reference
          best = diff;
          best_r = r;
        }
        ++cnt;
      }
    }
    if (cnt > 0) {
      score += sum / (double)cnt;
      if (top != best_r && !chosen) top = best_r;
    }
    top -= r0;
  }
  score_ret = score;
  top_ret = top;
}

__device__ __forceinline__ void discourse_score_entry(
    const int e, const int* __restrict__ spk, const int* __restrict__ rel_off, const int* __restrict__ rel_sense,
    const int* __restrict__ rel_conn, const double* __restrict__ rel_prom, int q_sense, int q_conn,
    int q_spk, double q_prom, double* __restrict__ score_out, int* __restrict__ top_out) {
  double sc;
  int tp;
  discourse_score_value(e, spk, rel_off, rel_sense, rel_conn, rel_prom, q_sense, q_conn, q_spk, q_prom, sc, tp);
  score_out[e] = sc;
  top_out[e] = tp;
}


__global__ void __launch_bounds__(256) discourse_scores_kernel(
    const int* __restrict__ spk, const int* __restrict__ rel_off, const int* __restrict__ rel_sense,
    const int* __restrict__ rel_conn, const double* __restrict__ rel_prom, int n_entries, int q_sense, int q_conn,
    int q_spk, double q_prom, double* __restrict__ score_out, int* __restrict__ top_out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_entries) return;
  discourse_score_entry(e, spk, rel_off, rel_sense, rel_conn, rel_prom, q_sense, q_conn, q_spk, q_prom, score_out, top_out);
}

// all queries of a batch in one launch: blockIdx.y = query; params[q] = (sense, connective, speaker, prominence) as
// doubles (the integer codes are exact), outputs [n_queries][n_entries]
__global__ void __launch_bounds__(256) discourse_scores_batched_kernel(
    const int* __restrict__ spk, const int* __restrict__ rel_off, const int* __restrict__ rel_sense,
    const int* __restrict__ rel_conn, const double* __restrict__ rel_prom, int n_entries,
    const double* __restrict__ params, double* __restrict__ score_out, int* __restrict__ top_out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_entries) return;
  const double* qp = params + 4 * blockIdx.y;
  const size_t o = (size_t)blockIdx.y * n_entries;
  discourse_score_entry(e, spk, rel_off, rel_sense, rel_conn, rel_prom, (int)qp[0], (int)qp[1], (int)qp[2], qp[3],
                        score_out + o, top_out + o);
}

// rag/gesture_type_retrieval.py:41-117 for one query label (type, word): per DB entry, over its non-beat labels of
// the query type: +2 type present, +2 same speaker, +5 exact word (top = first label with that word) or
// +3/(1+2*max similarity) (top = first label of maximal similarity); word codes index the similarity vector the
// host computed for this query word (get_word_similarity_score against every DB word), float64 like the
// reference's Python floats.  top_out = index within the entry's non-beat labels, -1 if the type is absent.
__global__ void __launch_bounds__(256) gesture_scores_kernel(const int* __restrict__ spk, const int* __restrict__ lab_off,
                                                            const int* __restrict__ lab_type, const int* __restrict__ lab_word,
                                                            const double* __restrict__ lab_prom,
                                                            const double* __restrict__ word_sim, int n_entries, int q_type,
                                                            int q_word, int q_spk, double spk_bonus, double q_prom,
                                                            int sim_f32, double* __restrict__ score_out,
                                                            int* __restrict__ top_out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_entries) return;
  const int r0 = lab_off[e], r1 = lab_off[e + 1];
  int exact = -1, best = -1;
  double best_sim = 0.0;
  bool any = false;
  for (int r = r0; r < r1; ++r) {
    if (lab_type[r] != q_type) continue;
    any = true;
    const int w = lab_word[r];
    if (exact < 0 && q_word >= 0 && w == q_word) exact = r;
    const double sv = word_sim[w];
    if (best < 0 || sv > best_sim) {   // np.argmax: first maximum
      best = r;
      best_sim = sv;
    }
  }
  double score = 0.0;
  int top = -1;
  if (any) {
    bool f32 = false;
    score += 2.0;
    if (spk[e] == q_spk) score += spk_bonus;
    if (exact >= 0) {
      score += 5.0;
      top = exact - r0;
    } else {
      // a similarity model that returns numpy float32 (gensim) makes the reference's score float32 from here on
      // under NumPy >= 2 (NEP 50: python scalar (+|*|/) np.float32 -> float32); python floats, and float32 scalars
      // under the reference's pinned NumPy < 1.24 (scalar-scalar promotion), keep it float64
      if (sim_f32) {
        const float s = (float)best_sim;
        score = (double)((float)score + 3.0f / (1.0f + 2.0f * s));
        f32 = true;
      } else {
        score += 3.0 / (1.0 + 2.0 * best_sim);
      }
      top = best - r0;
    }
    // llm method only (rag/llm_retrieval.py:385-417): mean of 4/(1+2|prominence difference|) over the labels of
    // the query type with a known prominence; the label of smallest difference (first minimum) is reported
    if (lab_prom != nullptr && q_prom == q_prom) {
      double sum = 0.0, bestd = 0.0;
      int cnt = 0, best_r = -1;
      for (int r = r0; r < r1; ++r) {
        if (lab_type[r] != q_type) continue;
        const double p = lab_prom[r];
        if (p != p) continue;
        const double diff = fabs(p - q_prom);
        sum += 4.0 / (1.0 + 2.0 * diff);
        if (best_r < 0 || diff < bestd) {
          bestd = diff;
          best_r = r;
        }
        ++cnt;
      }
      if (cnt > 0) {
        const double ps = sum / (double)cnt;
        score = f32 ? (double)((float)score + (float)ps) : score + ps;
        top = best_r - r0;
      }
    }
  }
  score_out[e] = score;
  top_out[e] = top;
}

__global__ void __launch_bounds__(256) text_diag_sim_kernel(const float* __restrict__ q, int Lq,
                                                           const float* __restrict__ feats,
                                                           const int64_t* __restrict__ feat_off,
                                                           const int* __restrict__ cand, int dim,
                                                           double* __restrict__ out) {
  __shared__ double red[4];
  const int c = cand[blockIdx.x];
  const int64_t o0 = feat_off[c];
  const int Ld = (int)(feat_off[c + 1] - o0);
  const int n = Lq < Ld ? Lq : Ld;
  const int d4 = dim >> 2;
  const float4* qa = reinterpret_cast<const float4*>(q);
  const float4* fa = reinterpret_cast<const float4*>(feats + o0 * dim);
  double acc = 0.0;
  for (int i = threadIdx.x; i < n * d4; i += 256) {
    const float4 a = qa[i], b = fa[i];
    acc += (double)(a.x * b.x) + (double)(a.y * b.y) + (double)(a.z * b.z) + (double)(a.w * b.w);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = n > 0 ? ((red[0] + red[1]) + (red[2] + red[3])) / (double)n : 0.0;
}

// ---- candidate selection on the device --------------------------------------------------------
// The reference sorts all N scores and walks the ranking until it has 10 entries, so only entries
// whose score reaches the 10th largest one (with multiplicity) can ever be visited -- including
// EVERY entry tied at that score (they are ordered by the text-similarity tie-break).  Instead of
// copying N scores to the host, find that threshold on the device and compact the survivors:
//   topk_local : each workgroup extracts the 10 largest scores of its slice (10 rounds of a
//                block-wide max that consumes one instance per round: multiplicity is preserved);
//   topk_merge : one workgroup repeats that over the per-block lists -> threshold;
//   compact    : entries with score >= threshold and score > 0 are appended (atomic cursor;
//                the host re-sorts the few survivors by index, which restores the stable order).
// Scores are >= 0, comparisons only: the selection is exact.
constexpr int SEL_K = 10;
constexpr int SEL_IPT = 4;   // items per thread held in registers

__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

// The 10 largest values (with multiplicity) of the 256 x SEL_IPT items a workgroup holds in registers; tops[] is uniform
// across the block on return.  Per wave: 10 rounds of "largest remaining value" on wave shuffles alone (the owner of a
// round's maximum -- the lowest lane that holds it -- consumes ONE instance); the four waves' lists meet in LDS and wave 0
// repeats the rounds over those 40 values.  One barrier pair per call (round 3 took two per round plus an atomic: the ten
// rounds of a slice cost more than its whole sweep).
__device__ __forceinline__ void wave_top_rounds(double (&v)[SEL_IPT], double (&tops)[SEL_K]) {
  const int lane = threadIdx.x & 63;
  for (int r = 0; r < SEL_K; ++r) {
    double m = v[0];
#pragma unroll
    for (int i = 1; i < SEL_IPT; ++i) m = v[i] > m ? v[i] : m;
    const double wm = wave_max_f64(m);
    const unsigned long long holders = __ballot(m == wm);
    if (lane == __ffsll((long long)holders) - 1) {   // consume ONE instance of the maximum
      bool done = false;
#pragma unroll
      for (int i = 0; i < SEL_IPT; ++i)
        if (!done && v[i] == wm) {
          v[i] = -1.0;
          done = true;
        }
    }
    tops[r] = wm;
  }
}

__device__ __forceinline__ void block_top_rounds(double (&v)[SEL_IPT], double* red, double (&tops)[SEL_K]) {
  __shared__ double lists[4 * SEL_K + SEL_K];
  (void)red;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  wave_top_rounds(v, tops);
  if (lane < SEL_K) {
    double t = -1.0;
    for (int r = 0; r < SEL_K; ++r) t = (lane == r) ? tops[r] : t;
    lists[wave * SEL_K + lane] = t;
  }
  __syncthreads();
  if (wave == 0) {
    double w[SEL_IPT];
#pragma unroll
    for (int i = 0; i < SEL_IPT; ++i) w[i] = -1.0;
    if (lane < 4 * SEL_K) w[0] = lists[lane];
    double t2[SEL_K];
    wave_top_rounds(w, t2);
    if (lane < SEL_K) {
      double t = -1.0;
      for (int r = 0; r < SEL_K; ++r) t = (lane == r) ? t2[r] : t;
      lists[4 * SEL_K + lane] = t;
    }
  }
  __syncthreads();
  for (int r = 0; r < SEL_K; ++r) tops[r] = lists[4 * SEL_K + r];
  __syncthreads();                                   // (the lists are rewritten by the caller's next pass)
}

// blockIdx.y = query of a batched selection (strides in elements; 0 for a single query)
__global__ void __launch_bounds__(256) topk_local_kernel(const double* __restrict__ score, int n,
                                                        double* __restrict__ block_tops, size_t score_stride,
                                                        size_t ws_stride) {
  __shared__ double red[4];
  score += blockIdx.y * score_stride;
  block_tops += blockIdx.y * ws_stride;
  const int per_block = 256 * SEL_IPT;
  double v[SEL_IPT], tops[SEL_K];
#pragma unroll
  for (int i = 0; i < SEL_IPT; ++i) {
    const int e = blockIdx.x * per_block + i * 256 + threadIdx.x;
    v[i] = e < n ? score[e] : -1.0;
  }
  block_top_rounds(v, red, tops);
  if (threadIdx.x == 0)
    for (int r = 0; r < SEL_K; ++r) block_tops[blockIdx.x * SEL_K + r] = tops[r];
}

__global__ void __launch_bounds__(256) topk_merge_kernel(const double* __restrict__ block_tops, int n_lists,
                                                        int n_entries, double* __restrict__ threshold,
                                                        int* __restrict__ cursor, size_t ws_stride) {
  __shared__ double red[4];
  block_tops += blockIdx.y * ws_stride;
  threshold += blockIdx.y * ws_stride;
  cursor += blockIdx.y;
  // the 10th largest score lies among the per-block top-10 lists; fold them 1024 at a time
  double carry[SEL_K];
  for (int r = 0; r < SEL_K; ++r) carry[r] = -1.0;
  const int total = n_lists * SEL_K;
  for (int base = 0; base < total; base += 256 * SEL_IPT - SEL_K) {
    double v[SEL_IPT], tops[SEL_K];
#pragma unroll
    for (int i = 0; i < SEL_IPT; ++i) {
      const int j = i * 256 + threadIdx.x;          // slots [0, SEL_K) carry the running top-10
      double x = -1.0;
      if (j >= SEL_K && base + j - SEL_K < total) x = block_tops[base + j - SEL_K];
      v[i] = x;
    }
    if (threadIdx.x < SEL_K) {
      double c = -1.0;
      for (int r = 0; r < SEL_K; ++r) c = ((int)threadIdx.x == r) ? carry[r] : c;
      v[0] = c;
    }
    block_top_rounds(v, red, tops);
    for (int r = 0; r < SEL_K; ++r) carry[r] = tops[r];   // tops[] is uniform across the block
  }
  if (threadIdx.x == 0) {
    // fewer than 10 entries in total, or a 10th largest of 0: keep every positive score
    double t = carry[SEL_K - 1];
    if (n_entries <= 64 || t < 0.0) t = 0.0;
    *threshold = t;
    *cursor = 0;
  }
}

__global__ void __launch_bounds__(256) compact_kernel(const double* __restrict__ score, const int* __restrict__ top,
                                                     int n, const double* __restrict__ threshold,
                                                     int* __restrict__ cursor, int cap, int* __restrict__ out_idx,
                                                     int* __restrict__ out_top, double* __restrict__ out_score,
                                                     size_t score_stride, size_t ws_stride) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  score += blockIdx.y * score_stride;
  top += blockIdx.y * score_stride;
  threshold += blockIdx.y * ws_stride;
  cursor += blockIdx.y;
  out_idx += (size_t)blockIdx.y * cap;
  out_top += (size_t)blockIdx.y * cap;
  out_score += (size_t)blockIdx.y * cap;
  const double s = score[e];
  if (s >= *threshold && s > 0.0) {
    const int pos = atomicAdd(cursor, 1);
    if (pos < cap) {
      out_idx[pos] = e;
      out_top[pos] = top[e];
      out_score[pos] = s;
    }
  }
}


// ---- sweep + selection fused (round 4): the scores never go to memory ---------------------------------------------------
// The batched sweep above writes a float64 score and an int32 relation index per (query, entry) -- 12 B x Q x N, 19 MB for
// the 48 query relations of a guided batch over 32 768 entries -- which three selection launches then read back twice.
// Only the 10th largest score of every query (the threshold) and the few entries that reach it are ever used.  Two launches:
//   sweep_tops    : grid (entry slices of 1024, queries): every thread scores its 4 entries in registers, the workgroup
//                   extracts the slice's 10 largest scores (block_top_rounds) -> block_tops[query][slice][10];
//                   the (0, query) workgroup also zeroes the query's cursor;
//   sweep_compact : same grid: every workgroup first folds the query's per-slice lists into the threshold (n_slices x 10
//                   values: one block_top_rounds pass per 1014 of them), then scores its 4 entries AGAIN (the integer-coded
//                   CSR is 1.2 MB: it stays in L2) and appends the survivors.
// Same arithmetic, same comparisons: identical survivors (order arbitrary as before; the host sorts them by index).
__global__ void __launch_bounds__(256) sweep_tops_kernel(
    const int* __restrict__ spk, const int* __restrict__ rel_off, const int* __restrict__ rel_sense,
    const int* __restrict__ rel_conn, const double* __restrict__ rel_prom, int n_entries,
    const double* __restrict__ params, double* __restrict__ block_tops, int* __restrict__ cursor, size_t ws_stride) {
  __shared__ double red[4];
  const double* qp = params + 4 * blockIdx.y;
  const int q_sense = (int)qp[0], q_conn = (int)qp[1], q_spk = (int)qp[2];
  const double q_prom = qp[3];
  double v[SEL_IPT], tops[SEL_K];
#pragma unroll
  for (int i = 0; i < SEL_IPT; ++i) {
    const int e = blockIdx.x * (256 * SEL_IPT) + i * 256 + threadIdx.x;
    v[i] = -1.0;
    if (e < n_entries) {
      int tp;
      discourse_score_value(e, spk, rel_off, rel_sense, rel_conn, rel_prom, q_sense, q_conn, q_spk, q_prom, v[i], tp);
    }
  }
  block_top_rounds(v, red, tops);
  if (threadIdx.x == 0) {
    double* bt = block_tops + blockIdx.y * ws_stride + blockIdx.x * SEL_K;
    for (int r = 0; r < SEL_K; ++r) bt[r] = tops[r];
    if (blockIdx.x == 0) cursor[blockIdx.y] = 0;
  }
}

__global__ void __launch_bounds__(256) sweep_compact_kernel(
    const int* __restrict__ spk, const int* __restrict__ rel_off, const int* __restrict__ rel_sense,
    const int* __restrict__ rel_conn, const double* __restrict__ rel_prom, int n_entries,
    const double* __restrict__ params, const double* __restrict__ block_tops, int n_lists, int* __restrict__ cursor,
    int cap, int* __restrict__ out_idx, int* __restrict__ out_top, double* __restrict__ out_score, size_t ws_stride) {
  __shared__ double red[4];
  block_tops += blockIdx.y * ws_stride;
  // threshold = 10th largest of the query's per-slice lists (as topk_merge_kernel)
  double carry[SEL_K];
  for (int r = 0; r < SEL_K; ++r) carry[r] = -1.0;
  const int total = n_lists * SEL_K;
  for (int base = 0; base < total; base += 256 * SEL_IPT - SEL_K) {
    double v[SEL_IPT], tops[SEL_K];
#pragma unroll
    for (int i = 0; i < SEL_IPT; ++i) {
      const int j = i * 256 + threadIdx.x;
      double x = -1.0;
      if (j >= SEL_K && base + j - SEL_K < total) x = block_tops[base + j - SEL_K];
      v[i] = x;
    }
    if (threadIdx.x < SEL_K) {
      double c = -1.0;
      for (int r = 0; r < SEL_K; ++r) c = ((int)threadIdx.x == r) ? carry[r] : c;
      v[0] = c;
    }
    block_top_rounds(v, red, tops);
    for (int r = 0; r < SEL_K; ++r) carry[r] = tops[r];
  }
  double thr = carry[SEL_K - 1];
  if (n_entries <= 64 || thr < 0.0) thr = 0.0;
  const double* qp = params + 4 * blockIdx.y;
  const int q_sense = (int)qp[0], q_conn = (int)qp[1], q_spk = (int)qp[2];
  const double q_prom = qp[3];
  cursor += blockIdx.y;
  out_idx += (size_t)blockIdx.y * cap;
  out_top += (size_t)blockIdx.y * cap;
  out_score += (size_t)blockIdx.y * cap;
#pragma unroll
  for (int i = 0; i < SEL_IPT; ++i) {
    const int e = blockIdx.x * (256 * SEL_IPT) + i * 256 + threadIdx.x;
    if (e >= n_entries) continue;
    double sc;
    int tp;
    discourse_score_value(e, spk, rel_off, rel_sense, rel_conn, rel_prom, q_sense, q_conn, q_spk, q_prom, sc, tp);
    if (sc >= thr && sc > 0.0) {
      const int pos = atomicAdd(cursor, 1);
      if (pos < cap) {
        out_idx[pos] = e;
        out_top[pos] = tp;
        out_score[pos] = sc;
      }
    }
  }
}


// ------------------------------------------------------------------------------ fuzzy word similarity
// fuzzywuzzy 0.18 `fuzz.partial_ratio` (pure-python flavour: difflib.SequenceMatcher) of one query string against every
// word of the DB vocabulary, one thread per word -- the similarity `get_word_similarity_score` effectively returns
// (rag/utils.py:239-272: the embedding models are undefined, every call ends in `fuzz.partial_ratio(w1, w2) / 100`).
// Strings are code-point arrays.  difflib's matcher for sequences below its autojunk length, written out:
//   find_longest_match: longest common substring of a[alo:ahi], b[blo:bhi]; among equals the earliest in a, then in b
//   get_matching_blocks: LIFO queue of the pieces left and right of each match
//   ratio = 2 * matched / (len a + len b)
// partial_ratio: for every diagonal d = max(j - i, 0) of the outer blocks (the (la, lb, 0) sentinel included) the ratio
// of `shorter` against longer[d : d + la]; > .995 -> 100; else int(round(100 * max)) (round half to even, like Python).
constexpr int PR_MAX = 48;   // longest string handled on the device (longer: out = NaN, the host computes it)

struct PrQuery {
  int len;
  int c[PR_MAX];
};

__device__ int pr_longest(const int* a, const int* b, int alo, int ahi, int blo, int bhi, int& bi, int& bj) {
  unsigned char prev[PR_MAX], cur[PR_MAX];
  for (int j = blo; j < bhi; ++j) prev[j] = 0;
  int best = 0;
  bi = alo;
  bj = blo;
  for (int i = alo; i < ahi; ++i) {
    const int ai = a[i];
    for (int j = blo; j < bhi; ++j) {
      int k = 0;
      if (b[j] == ai) {
        k = (j > blo ? prev[j - 1] : 0) + 1;
        if (k > best) {
          best = k;
          bi = i - k + 1;
          bj = j - k + 1;
        }
      }
      cur[j] = (unsigned char)k;
    }
    for (int j = blo; j < bhi; ++j) prev[j] = cur[j];
  }
  return best;
}

// matched characters of SequenceMatcher(None, a, b); diag (optional): bit d set for every block diagonal max(j - i, 0)
__device__ int pr_matches(const int* a, int la, const int* b, int lb, unsigned long long* diag) {
  short st[PR_MAX + 2][4];
  int sp = 0, total = 0;
  st[0][0] = 0; st[0][1] = (short)la; st[0][2] = 0; st[0][3] = (short)lb;
  sp = 1;
  while (sp > 0) {
    --sp;
    const int alo = st[sp][0], ahi = st[sp][1], blo = st[sp][2], bhi = st[sp][3];
    int i, j;
    const int k = pr_longest(a, b, alo, ahi, blo, bhi, i, j);
    if (k) {
      total += k;
      if (diag) *diag |= 1ull << (j - i > 0 ? j - i : 0);
      if (alo < i && blo < j) {
        st[sp][0] = (short)alo; st[sp][1] = (short)i; st[sp][2] = (short)blo; st[sp][3] = (short)j;
        ++sp;
      }
      if (i + k < ahi && j + k < bhi) {
        st[sp][0] = (short)(i + k); st[sp][1] = (short)ahi; st[sp][2] = (short)(j + k); st[sp][3] = (short)bhi;
        ++sp;
      }
    }
  }
  return total;
}

__global__ void __launch_bounds__(64) partial_ratio_kernel(const int* __restrict__ vocab, const int* __restrict__ vlen,
                                                           int n_words, int stride, const PrQuery q,
                                                           double* __restrict__ out) {
  const int v = blockIdx.x * 64 + threadIdx.x;
  if (v >= n_words) return;
  const int l1 = vlen[v], l2 = q.len;
  if (l1 > PR_MAX || l2 > PR_MAX) {
    out[v] = __longlong_as_double(0x7ff8000000000000ll);   // NaN: too long for the device form
    return;
  }
  int s1[PR_MAX], s2[PR_MAX];
  for (int i = 0; i < l1; ++i) s1[i] = vocab[(size_t)v * stride + i];
  for (int i = 0; i < l2; ++i) s2[i] = q.c[i];
  bool same = l1 == l2;
  for (int i = 0; same && i < l1; ++i) same = s1[i] == s2[i];
  if (same) { out[v] = 1.0; return; }             // check_for_equivalence (also two empty strings)
  if (l1 == 0 || l2 == 0) { out[v] = 0.0; return; }
  const int* a = l1 <= l2 ? s1 : s2;               // shorter (ties: the first argument = the DB word)
  const int* b = l1 <= l2 ? s2 : s1;
  const int la = l1 <= l2 ? l1 : l2, lb = l1 <= l2 ? l2 : l1;
  unsigned long long diag = 1ull << (lb - la);     // the (la, lb, 0) sentinel block
  pr_matches(a, la, b, lb, &diag);
  double best = 0.0;
  bool hit = false;
  for (int d = 0; d <= lb && !hit; ++d) {
    if (!((diag >> d) & 1ull)) continue;
    const int ls = (d + la <= lb) ? la : lb - d;   // longer[d : d + la] is cut at the end of the string
    const int m = pr_matches(a, la, b + d, ls, nullptr);
    const double r = (la + ls) ? 2.0 * (double)m / (double)(la + ls) : 1.0;
    if (r > .995) hit = true;
    best = r > best ? r : best;
  }
  out[v] = hit ? 1.0 : rint(100.0 * best) / 100.0;
}
}  // namespace

static int select_launch(rg_handle* h, const double* score, const int* top, int n_entries, int n_queries,
                         double* workspace, int* cursor, int cap, int* out_idx, int* out_top, double* out_score,
                         void* stream) {
  const int nb = (n_entries + 256 * SEL_IPT - 1) / (256 * SEL_IPT);
  const size_t ws_stride = (size_t)nb * SEL_K + 1, sc_stride = (size_t)n_entries;
  hipStream_t s = rg_stream(stream);
  // workspace per query: [nb * 10] per-block tops, then the threshold
  hipLaunchKernelGGL(topk_local_kernel, dim3(nb, n_queries), dim3(256), 0, s, score, n_entries, workspace, sc_stride,
                     ws_stride);
  hipLaunchKernelGGL(topk_merge_kernel, dim3(1, n_queries), dim3(256), 0, s, workspace, nb, n_entries,
                     workspace + nb * SEL_K, cursor, ws_stride);
  hipLaunchKernelGGL(compact_kernel, dim3((n_entries + 255) / 256, n_queries), dim3(256), 0, s, score, top, n_entries,
                     workspace + nb * SEL_K, cursor, cap, out_idx, out_top, out_score, sc_stride, ws_stride);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_select_top_scores(rg_handle* h, const double* score, const int* top, int n_entries,
                                    double* workspace, int* cursor, int cap, int* out_idx, int* out_top,
                                    double* out_score, void* stream) {
  RG_REQUIRE(h, score && top && workspace && cursor && out_idx && out_top && out_score, "null pointer");
  RG_REQUIRE(h, n_entries > 0 && cap > 0, "bad shape");
  return select_launch(h, score, top, n_entries, 1, workspace, cursor, cap, out_idx, out_top, out_score, stream);
}

extern "C" int rg_select_top_scores_batched(rg_handle* h, const double* score, const int* top, int n_entries,
                                            int n_queries, double* workspace, int* cursor, int cap, int* out_idx,
                                            int* out_top, double* out_score, void* stream) {
  RG_REQUIRE(h, score && top && workspace && cursor && out_idx && out_top && out_score, "null pointer");
  RG_REQUIRE(h, n_entries > 0 && cap > 0 && n_queries > 0 && n_queries <= 65535, "bad shape");
  return select_launch(h, score, top, n_entries, n_queries, workspace, cursor, cap, out_idx, out_top, out_score, stream);
}

extern "C" int rg_discourse_select_fused(rg_handle* h, const int* spk, const int* rel_off, const int* rel_sense,
                                        const int* rel_conn, const double* rel_prom, int n_entries, const double* params,
                                        int n_queries, double* workspace, int* cursor, int cap, int* out_idx, int* out_top,
                                        double* out_score, void* stream) {
  RG_REQUIRE(h, spk && rel_off && rel_sense && rel_conn && rel_prom && params, "null pointer");
  RG_REQUIRE(h, workspace && cursor && out_idx && out_top && out_score, "null pointer");
  RG_REQUIRE(h, n_entries > 0 && cap > 0 && n_queries > 0 && n_queries <= 65535, "bad shape");
  const int nb = (n_entries + 256 * SEL_IPT - 1) / (256 * SEL_IPT);
  const size_t ws_stride = (size_t)nb * SEL_K + 1;
  hipStream_t s = rg_stream(stream);
  hipLaunchKernelGGL(sweep_tops_kernel, dim3(nb, n_queries), dim3(256), 0, s, spk, rel_off, rel_sense, rel_conn, rel_prom,
                     n_entries, params, workspace, cursor, ws_stride);
  hipLaunchKernelGGL(sweep_compact_kernel, dim3(nb, n_queries), dim3(256), 0, s, spk, rel_off, rel_sense, rel_conn, rel_prom,
                     n_entries, params, workspace, nb, cursor, cap, out_idx, out_top, out_score, ws_stride);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_select_workspace_doubles(int n_entries) {
  return (n_entries + 256 * SEL_IPT - 1) / (256 * SEL_IPT) * SEL_K + 1;
}

extern "C" int rg_discourse_scores(rg_handle* h, const int* spk, const int* rel_off, const int* rel_sense,
                                   const int* rel_conn, const double* rel_prom, int n_entries, int q_sense,
                                   int q_conn, int q_spk, double q_prom, double* score_out, int* top_out,
                                   void* stream) {
  RG_REQUIRE(h, spk && rel_off && rel_sense && rel_conn && rel_prom && score_out && top_out, "null pointer");
  RG_REQUIRE(h, n_entries > 0, "empty database");
  hipLaunchKernelGGL(discourse_scores_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, rg_stream(stream), spk,
                     rel_off, rel_sense, rel_conn, rel_prom, n_entries, q_sense, q_conn, q_spk, q_prom, score_out,
                     top_out);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_discourse_scores_batched(rg_handle* h, const int* spk, const int* rel_off, const int* rel_sense,
                                           const int* rel_conn, const double* rel_prom, int n_entries,
                                           const double* params, int n_queries, double* score_out, int* top_out,
                                           void* stream) {
  RG_REQUIRE(h, spk && rel_off && rel_sense && rel_conn && rel_prom && params && score_out && top_out, "null pointer");
  RG_REQUIRE(h, n_entries > 0 && n_queries > 0 && n_queries <= 65535, "bad shape");
  hipLaunchKernelGGL(discourse_scores_batched_kernel, dim3((n_entries + 255) / 256, n_queries), dim3(256), 0,
                     rg_stream(stream), spk, rel_off, rel_sense, rel_conn, rel_prom, n_entries, params, score_out, top_out);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_text_diag_sim(rg_handle* h, const float* q, int Lq, const float* feats, const int64_t* feat_off,
                                const int* cand, int n_cand, int dim, double* out, void* stream) {
  RG_REQUIRE(h, q && feats && feat_off && cand && out, "null pointer");
  RG_REQUIRE(h, n_cand > 0 && Lq > 0 && dim % 4 == 0, "bad shape");
  hipLaunchKernelGGL(text_diag_sim_kernel, dim3(n_cand), dim3(256), 0, rg_stream(stream), q, Lq, feats, feat_off,
                     cand, dim, out);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_gesture_scores(rg_handle* h, const int* spk, const int* lab_off, const int* lab_type, const int* lab_word,
                                 const double* lab_prom, const double* word_sim, int n_entries, int q_type, int q_word,
                                 int q_spk, double spk_bonus, double q_prom, int sim_f32, double* score_out, int* top_out,
                                 void* stream) {
  RG_REQUIRE(h, spk && lab_off && lab_type && lab_word && word_sim && score_out && top_out, "null pointer");
  RG_REQUIRE(h, n_entries > 0, "empty database");
  hipLaunchKernelGGL(gesture_scores_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, rg_stream(stream), spk, lab_off,
                     lab_type, lab_word, lab_prom, word_sim, n_entries, q_type, q_word, q_spk, spk_bonus, q_prom, sim_f32,
                     score_out, top_out);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_partial_ratio(rg_handle* h, const int* vocab, const int* vocab_len, int n_words, int stride,
                                const int* query_host, int query_len, double* out, void* stream) {
  RG_REQUIRE(h, vocab && vocab_len && out && (query_host || query_len == 0), "null pointer");
  RG_REQUIRE(h, n_words > 0 && stride >= 1 && query_len >= 0, "bad shape");
  PrQuery q;
  q.len = query_len;
  for (int i = 0; i < PR_MAX; ++i) q.c[i] = (i < query_len && query_len <= PR_MAX) ? query_host[i] : 0;
  hipLaunchKernelGGL(partial_ratio_kernel, dim3((n_words + 63) / 64), dim3(64), 0, rg_stream(stream), vocab, vocab_len,
                     n_words, stride, q, out);
  RG_CHECK_LAUNCH(h);
  return RG_OK;
}

extern "C" int rg_partial_ratio_max_len(void) { return PR_MAX; }
