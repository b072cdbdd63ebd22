#!/bin/bash
# PMC counter passes (rocprofv3 --pmc, kernel-trace only) over a short command; one pass per counter group.
# Usage: tools/gpu_pmc.sh <tag> <command...>      output: gpurun_out/pmc_<tag>_<n>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
CGROUPS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
 "GRBM_GUI_ACTIVE FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
n=0
for g in "${CGROUPS[@]}"; do
  out="$ROOT/gpurun_out/pmc_${TAG}_$n"
  rm -rf "$out"
  (cd /tmp && timeout 600 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o pmc -- "$@" > "$out.log" 2>&1)
  echo "pmc group $n rc=$?" >> "$out.log"
  n=$((n+1))
done
ls gpurun_out
