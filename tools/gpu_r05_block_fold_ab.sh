#!/bin/bash
# r05 run b: the folded-norm kernels (tests), parity of the folded forward at S = 4096, block A/B folded / unfolded on ONE box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x --tb=short -p no:cacheprovider -k "rowss or scaled_epilogues or gemm_rope or gemm_swiglu" > $O/r05_fold_kernels.log 2>&1; echo "kernels rc=$?" >> $O/r05_fold_kernels.log; tail -4 $O/r05_fold_kernels.log
timeout 900 python -m pytest tests/test_parity_long_gpu.py -q -x --tb=short -p no:cacheprovider -s -k "folded_norms or (benchmarked_length and 4096 and bf16)" > $O/r05_fold_parity.log 2>&1; echo "parity rc=$?" >> $O/r05_fold_parity.log; grep -E "S=4096|passed|failed|rc=" $O/r05_fold_parity.log | tail -6
for i in 1 2 3; do
  timeout 300 python bench.py --mode block --block-unfolded 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['block']; print('unfolded', round(b['ms_per_block'],4), round(b['roofline']['frac'],4), {k:round(v['us_per_call'],1) for k,v in b['kernels'].items()})" >> $O/r05_fold_block_ab.txt
  timeout 300 python bench.py --mode block 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['block']; print('folded  ', round(b['ms_per_block'],4), round(b['roofline']['frac'],4), {k:round(v['us_per_call'],1) for k,v in b['kernels'].items()})" >> $O/r05_fold_block_ab.txt
done
cat $O/r05_fold_block_ab.txt
