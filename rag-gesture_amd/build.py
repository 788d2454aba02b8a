"""Build the C-ABI HIP extension (librg_gesture.so) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so (git-ignored) travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
DIAG = os.environ.get("RG_DIAG") == "1"   # diagnostic build with in-kernel phase stamps (never the product)
TAG = "diag" if DIAG else os.environ.get("RG_LIB_TAG", "")   # RG_LIB_TAG=x: an experiment build (RG_EXTRA_FLAGS) beside the product library
LIB_PATH = os.path.join(PKG_DIR, "librg_gesture%s.so" % ("_" + TAG if TAG else ""))
OBJ_DIR = os.path.join(PKG_DIR, "csrc", "_obj" + ("_" + TAG if TAG else ""))
ARCH = "gfx950"
# NO_PACKED_FP32: the device code is compiled WITHOUT packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 /
# v_pk_fma_f32, which the compiler forms from pairs of independent fp32 operations on gfx90a and later).
# Measured on MI355X with ROCm 7.2 (NOTEBOOK section 9, profiles/r04d_packed_fp32_hazard.txt): the HIGH half of a
# v_pk_add_f32 result, read by the next instructions, was occasionally STALE in lanes 48-63 -- the last of the four 16-lane
# passes -- when the SIMD was shared with waves of other kernels (other streams): in rg_6d_to_aa one term of a quaternion
# came out as the partial sum the register held before, i.e. 16 consecutive joints of a decoded pose wrong about once per
# 10^4 launches.  That was the second source of the round-3 "flaky" bit mismatches (the other one was the shared decode
# graph): 12-37 mismatching batches per 200 pipelined passes with packed ops, 0 per 200 without (same box, same script:
# profiles/dbg/hands_flake_probe.py); the GELU epilogue of rg_gemm showed the same signature once in 70 test runs.
# More wait states behind the producers (s_nop patched into the assembly: behind transcendentals, in front of lane-mask
# readers) did NOT help; not forming the packed instructions does.  Cost: see NOTEBOOK section 9 (rg_seq launch time).
# Round 5: a stand-alone reproducer (profiles/dbg/pk_f32_repro.hip: the rg_6d_to_aa arithmetic WITH packed instructions beside up
# to 20 busy streams, 6 x 10^5 launches, profiles/r05s_pk_f32_repro.txt) did NOT show the effect: a hardware hazard is not
# established, the mechanism behind the product-side A/B above is unknown, and the switch stays as a mitigation.  Building only
# the two denoiser kernels WITH packed fp32 (their workgroups own a whole compute unit: nothing shares their SIMDs) was measured
# too: bit-identical, race_stress 40 / 40, and worth 0.5 % of a launch (2 574 vs 2 585 us) -- not taken.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# Round 6: the effect above followed SIMD SHARING, not (only) packed arithmetic -- rg_seq2_kernel without a single v_pk_*_f32
# failed the same way as soon as its register count left room for foreign waves on its SIMDs (csrc/rg_common.h
# RG_OWN_THE_SIMD, NOTEBOOK 11.3).  The units below hold only kernels that own their SIMDs (256 registers per wave, two waves
# per SIMD); their epilogues are written two-wide (rg_common.h rg_fma2 ...) and get the packed instructions.  Results do not
# depend on the switch (every packed operation is the same IEEE operation per element).
PACKED_FP32_UNITS = set(os.environ.get("RG_PACKED_UNITS", "rg_seq.hip rg_seq2.hip rg_seqx.hip rg_venc.hip rg_vdec.hip rg_condkv.hip").split())
BASE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
              "-mllvm", "-amdgpu-early-inline-all=true"]
FLAGS = BASE_FLAGS + NO_PACKED_FP32 + os.environ.get("RG_EXTRA_FLAGS", "").split()


def flags_for(src):
    if os.path.basename(src) in PACKED_FP32_UNITS:
        return BASE_FLAGS + os.environ.get("RG_EXTRA_FLAGS", "").split()
    return FLAGS


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "rg_gesture.h"))
    return max(os.path.getmtime(p) for p in hdrs)


def _compile(src, hdr_mtime, force):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    cmd = [_hipcc()] + flags_for(src) + (["-DRG_STAMPS"] if DIAG else []) + ["-c", src, "-o", obj]
    stamp, want = obj + ".cmd", " ".join(cmd)
    same_cmd = os.path.exists(stamp) and open(stamp).read() == want      # (a changed flag set rebuilds, not only a changed source)
    if (not force and same_cmd and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= hdr_mtime):
        return obj
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(want)
    # (the host pass of hipcc sees the device-only feature switch too and says so: not a problem of this build)
    err = "\n".join(l for l in r.stderr.splitlines() if "packed-fp32-ops" not in l)
    if err.strip():
        sys.stderr.write(err + "\n")
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    hm = _deps_mtime()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hm, force), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
