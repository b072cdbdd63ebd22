#!/bin/bash
# session 2, run 2: where the forward's time goes (ablations of the third form), S = 4096 only
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python tools/bench_attn_forms.py > $O/s2_2_attn_forms_ablation.txt 2>&1
grep -v "S=2048" $O/s2_2_attn_forms_ablation.txt | tail -40
