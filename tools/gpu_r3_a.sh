#!/bin/bash
# r03 pass A: the K-step-64 main loop of gemm_pp256_kernel against the K-step-32 one (parity, per-shape rates, ablations,
# block benchmark), and the vendor library's kernels on the same shapes (names + rates) for comparison.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x --timeout 300 -p no:cacheprovider -k "gemm and not skinny" > $O/a_kernels.log 2>&1
echo "kernels rc=$?" >> $O/a_kernels.log
MH_BENCH_SHAPES=nn timeout 600 python tools/bench_gemm.py 1,2,1,2 > $O/a_bench_gemm_nn.log 2>&1
MH_BENCH_SHAPES=nnq timeout 300 python tools/bench_gemm.py 2,12,32,42,52 > $O/a_bench_gemm_abl.log 2>&1
timeout 300 python tools/bench_hipblaslt_torch.py > $O/a_hipblaslt.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/a_prof_blaslt -o blaslt -- python $GRAFT_REPO_ROOT/tools/bench_hipblaslt_torch.py > $GRAFT_REPO_ROOT/$O/a_prof_blaslt.log 2>&1)
tools/gpu_block_ab.sh MH_GEMM 1 2 2 > /dev/null 2>&1
cp $O/block_ab.txt $O/a_block_ab.txt
tail -3 $O/a_kernels.log
grep -E "variant" $O/a_bench_gemm_nn.log | tail -4
cat $O/a_block_ab.txt
