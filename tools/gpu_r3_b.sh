#!/bin/bash
# r03 pass B: the three schedules of the K-step-64 main loop (variants 2, 3, 4) against the K-step-32 loop: parity,
# rates, ablations, in-kernel timelines, block benchmark.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x --timeout 300 -p no:cacheprovider -k "gemm and not skinny" > $O/b_kernels.log 2>&1
echo "kernels rc=$?" >> $O/b_kernels.log
MH_BENCH_SHAPES=nnq timeout 600 python tools/bench_gemm.py 1,2,3,4,1,2,3,4 > $O/b_bench_gemm.log 2>&1
MH_BENCH_SHAPES=nnq timeout 300 python tools/bench_gemm.py 13,33,43,53,14,34,44,54 > $O/b_bench_gemm_abl.log 2>&1
timeout 300 python tools/gemm_timeline.py 2,3,4 32768 1024 4096 > $O/b_timeline_k4096.log 2>&1
timeout 300 python tools/gemm_timeline.py 3,4 32768 8192 1024 > $O/b_timeline_k1024.log 2>&1
for i in 1 2; do for v in 1 3 4; do
  MH_GEMM=$v timeout 300 python bench.py --mode block 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('block',d)
print('MH_GEMM=$v', round(b['ms_per_block'],3), 'ms frac', round(b['roofline']['frac'],4), {k:round(x['us_per_call'],1) for k,x in b['kernels'].items()})"
done; done > $O/b_block_ab.txt 2>&1
tail -3 $O/b_kernels.log
grep -v amdgpu $O/b_bench_gemm.log | tail -30
cat $O/b_block_ab.txt
