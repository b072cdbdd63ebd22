#!/bin/bash
# r03 pass D: schedule 4 (one region per wave per slot) against schedule 2 and the K-step-32 loop; timelines; LDS counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$(pwd)
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x --timeout 300 -p no:cacheprovider -k "gemm and not skinny" > $O/d_kernels.log 2>&1
echo "kernels rc=$?" >> $O/d_kernels.log
MH_BENCH_SHAPES=nnq timeout 600 python tools/bench_gemm.py 1,4,6,1,4,6 > $O/d_bench_gemm.log 2>&1
timeout 300 python tools/gemm_timeline.py 4,6 32768 1024 4096 > $O/d_timeline_k4096.log 2>&1
for i in 1 2; do for v in 1 4 6; do
  MH_GEMM=$v timeout 300 python bench.py --mode block 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('block',d)
print('MH_GEMM=$v', round(b['ms_per_block'],3), 'ms frac', round(b['roofline']['frac'],4), {k:round(x['us_per_call'],1) for k,x in b['kernels'].items()})"
done; done > $O/d_block_ab.txt 2>&1
MH_BENCH_SHAPES=nnq tools/gpu_pmc.sh d_gemm python $R/tools/bench_gemm.py 1,4,6 > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_d_gemm_0 $O/pmc_d_gemm_1 $O/pmc_d_gemm_2 $O/pmc_d_gemm_3 > $O/d_pmc_summary.txt 2>&1
rm -rf $O/pmc_d_gemm_*/
tail -3 $O/d_kernels.log
grep -v amdgpu $O/d_bench_gemm.log | tail -24
cat $O/d_block_ab.txt
