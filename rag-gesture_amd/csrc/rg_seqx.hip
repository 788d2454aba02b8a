// One launch, two forms of the denoiser forward, chosen ON THE DEVICE when the launch starts (round 6).
//
// rg_seq2_forward (two sequences of a kind per workgroup: 64 workgroups x 1.37 ms for the pipeline's 128-sequence launches)
// costs 0.66 of the CU time of rg_seq_forward (one sequence per workgroup: 128 x 0.85 ms) and takes 1.6x as long.  With every
// lane of the pipeline busy the chip is full and CU time is what counts; while a pipeline fills or drains -- the first and
// the last rotation of a finite run, any gap in the traffic -- compute units idle and the latency of the 50 dependent
// launches of a chain is what counts.  The host cannot choose: chains are queued (as HIP graphs) up to 100 ms before they
// run, and it learns that no more work follows (flush()) after the last chains have been queued.  So the choice is made where
// the load is known: every lane publishes the workgroups its current launch holds (rg_lane_form, a one-thread kernel in front
// of each forward, in the same graph), takes the wide form when the other lanes' workgroups + its own wide launch fit the
// chip, and this kernel -- launched with the wide form's grid -- runs whichever form the lane's flag names; in the narrow
// form the surplus workgroups leave at once.  Both forms give the same bits (tests/test_denoiser_gpu.py), so the result of
// a batch never depends on the load.
// Bodies: csrc/rg_seq.hip (run_sequence, seq_block), csrc/rg_seq2.hip (run_pair, seq2_block), compiled here once more.
#define RG_SEQ_BODY_ONLY
#include "rg_common.h"
#include "rg_tail.h"      // (before the bodies: they include it inside their namespaces)
#include <type_traits>

namespace rgx_two {
#include "rg_seq2.hip"
}
namespace rgx_one {
#include "rg_seq.hip"
}

namespace {
constexpr int NTH_X = 512;
constexpr int LDS_X = rgx_one::LDS_BYTES > rgx_two::LDS_BYTES ? rgx_one::LDS_BYTES : rgx_two::LDS_BYTES;
static_assert(rgx_one::NTH == NTH_X && rgx_two::NTH == NTH_X, "eight waves per workgroup in both forms");

__global__ void __launch_bounds__(NTH_X) rg_seqx_kernel(const rg_seq_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RG_OWN_THE_SIMD();
  // Written by rg_lane_form_kernel earlier on this stream.  EVERY workgroup of the launch must read the same value, on
  // whichever XCD it runs and however late it starts: an agent-scope atomic load (served coherently, not from a line this
  // XCD's L2 may have kept from an earlier launch), of a word that only this lane's own arbitration kernel writes and that
  // shares its 128-byte line with nothing another stream's kernels write (rg_lane_form: one line per lane).
  const int wide = a.form ? __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.form, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
  if (wide) {
    if ((int)blockIdx.x < 2 * a.B) rgx_one::seq_block(a, blockIdx.x, 0, smem);
  } else {
    if ((int)blockIdx.x < rgx_two::seq2_grid(a.B, a.split, a.pairs)) rgx_two::seq2_block(a, blockIdx.x, smem);
  }
}

}  // namespace

extern "C" int rg_seqx_forward(rg_handle* h, const rg_seq_args* args_host, void* stream) {
  RG_REQUIRE(h, args_host, "null args");
  const rg_seq_args& a = *args_host;
  RG_REQUIRE(h, a.wstream && a.pstream && a.ustream && a.afrag && a.x && a.tbias && a.src_mask && a.qmask && a.head, "null pointer");
  RG_REQUIRE(h, a.xbuf && a.gbuf, "the two-sequence form needs its scratch buffers xbuf and gbuf");
  RG_REQUIRE(h, a.form, "form: the lane's flag (rg_lane_form) must be given");
  RG_REQUIRE(h, a.L >= 1 && a.L <= 8 && a.B >= 1 && a.T >= 1 && a.T <= rgx_two::TP, "unsupported shape (T <= 48, L <= 8)");
  RG_REQUIRE(h, a.step >= 0 && a.step < a.S && a.step_b >= 0 && a.step_b < a.S, "step out of range");
  RG_REQUIRE(h, a.dump_stage == 0, "diagnostic dumps: use rg_seq_forward / rg_seq2_forward");
  RG_REQUIRE(h, a.pairs == 0 || a.pairs == 1, "pairs must be 0 or 1");
  RG_REQUIRE(h, rg_tail::args_ok(a), "glue_ctr: glue must cover the B clips (n_a + n_b == B, T, D = 512), its pointers set, no dump");
  static rg_attr_once lds_once;
  if (!rg_reserve_lds(lds_once, rg_seqx_kernel, LDS_X)) {
    h->err = "rg_seqx_forward: cannot reserve LDS";
    return RG_ERR_HIP;
  }
  rg_prof_rec rec;
  if (h->profiling) {   // bench.py roofline: HIP events around the launch (variant 3), algorithmic FLOPs of the T token rows
    auto get_ev = [&]() {
      hipEvent_t e;
      if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); } else { (void)hipEventCreate(&e); }
      return e;
    };
    rec.start = get_ev(); rec.stop = get_ev();
    rec.variant = 3;
    const double unit = 2.0 * a.T * 512 * 512, att = 2.0 * a.T * 32 * 32 * 16;
    const double cond = (16 * a.L + 2) * unit + a.L * (2 + 3) * att, unc = (10 * a.L + 2) * unit + a.L * 2 * att;
    rec.flops = a.B * (cond + unc);
    (void)hipEventRecord(rec.start, rg_stream(stream));
  }
  const int g2 = rgx_two::seq2_grid(a.B, a.split, a.pairs);
  hipLaunchKernelGGL(rg_seqx_kernel, dim3(2 * a.B > g2 ? 2 * a.B : g2), dim3(NTH_X), LDS_X, rg_stream(stream), a);
  RG_CHECK_LAUNCH(h);
  if (h->profiling) {
    (void)hipEventRecord(rec.stop, rg_stream(stream));
    h->prof.push_back(rec);
  }
  return RG_OK;
}
