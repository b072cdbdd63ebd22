#!/bin/bash
# r03 pass E: the two-phase K-step-64 loop (variant 2) and its A/B builds (3: no setprio, 4: LDS-DMA ahead of the reads, 5: both);
# all-wave timelines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x --timeout 300 -p no:cacheprovider -k "gemm and not skinny" > $O/f_kernels.log 2>&1
echo "kernels rc=$?" >> $O/f_kernels.log
MH_BENCH_SHAPES=nnq timeout 600 python tools/bench_gemm.py 1,2,3,4,5,1,2,3,4,5 > $O/f_bench_gemm.log 2>&1
timeout 300 python tools/gemm_timeline.py 2,4 32768 1024 4096 > $O/f_timelinf_k4096.log 2>&1
tail -3 $O/f_kernels.log
grep -v amdgpu $O/f_bench_gemm.log | tail -36
