#!/bin/bash
# session 2, run 3: PMC passes over the attention kernels (third form), S = 4096
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MH_RUN_BWD=1
bash tools/gpu_pmc.sh attn_v3 python $(pwd)/tools/run_attn_once.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_attn_v3_* > gpurun_out/s2_3_pmc_attn_v3.txt 2>&1
cat gpurun_out/s2_3_pmc_attn_v3.txt | cut -c1-900
