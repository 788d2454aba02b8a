#!/bin/bash
# Static instruction histogram of one translation unit's device code, built with the product flags:
#   profiles/dbg/isa_hist.sh rg_seq2 [extra hipcc flags]   ->  /tmp/isa/<name>.s + counts by class
set -e
N=$1; shift
mkdir -p /tmp/isa
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-early-inline-all=true \
  -Xclang -target-feature -Xclang -packed-fp32-ops "$@" --cuda-device-only -S -o /tmp/isa/$N.s \
  /root/repo/rag-gesture_amd/csrc/$N.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|Function Name" | sed 's/.*remark: //'
grep -E "codeLenInByte" /tmp/isa/$N.s
awk '/^\t[a-z]/ {op=$1;
  if (op ~ /^v_mfma/) c["mfma"]++;
  else if (op ~ /^v_(exp|rcp|rsq|log|sqrt|sin|cos)/) c["trans"]++;
  else if (op ~ /^v_(readlane|writelane|readfirstlane)/) c["lane"]++;
  else if (op ~ /^v_/) c["valu"]++;
  else if (op ~ /^s_waitcnt/) c["waitcnt"]++;
  else if (op ~ /^s_barrier/) c["barrier"]++;
  else if (op ~ /^s_nop/) c["nop"]++;
  else if (op ~ /^s_/) c["salu"]++;
  else if (op ~ /^ds_/) c["lds"]++;
  else if (op ~ /^(buffer|global|flat)_/) c["vmem"]++;
  else c["other"]++;}
  END {for (k in c) printf "%s %d\n", k, c[k]}' /tmp/isa/$N.s | sort
