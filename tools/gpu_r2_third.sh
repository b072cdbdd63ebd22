#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_decode_gpu.py "tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length" "tests/test_kernels_gpu.py::test_sample_top_p_k_fused" "tests/test_kernels_gpu.py::test_gemm_skinny" tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider -s -x > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline > $O/gen.log 2> $O/gen.err
echo "gen rc=$?" >> $O/gen.err
timeout 300 python tools/decode_probe.py 2 > $O/decode_probe2.txt 2>&1
tools/gpu_pmc.sh block python $R/bench.py --mode block --steps 10 --warmup 3 > $O/pmc_block.log 2>&1
python tools/pmc_summary.py $O/pmc_block_0 $O/pmc_block_1 $O/pmc_block_2 $O/pmc_block_3 > $O/pmc_block_summary.txt 2>&1
rm -rf $O/pmc_block_0 $O/pmc_block_1 $O/pmc_block_2 $O/pmc_block_3
tail -n 12 $O/retest.log; tail -c 1200 $O/gen.log; tail -3 $O/gen.err; cat $O/decode_probe2.txt | tail -5; cat $O/pmc_block_summary.txt | cut -c1-1200
