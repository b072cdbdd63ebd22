#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest "tests/test_parity_long_gpu.py::test_training_step_gradients_at_S2048" "tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length" tests/test_decode_gpu.py::test_checkpoint_round_trips_on_device -q -m gpu --tb=short -p no:cacheprovider -s > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
timeout 300 tools/bin/skinny_probe > $O/skinny_probe.txt 2>&1
echo "probe rc=$?" >> $O/skinny_probe.txt
tools/gpu_pmc.sh block python bench.py --mode block --steps 10 --warmup 3 > $O/pmc_block.log 2>&1
python tools/pmc_summary.py $O/pmc_block_0 $O/pmc_block_1 $O/pmc_block_2 $O/pmc_block_3 > $O/pmc_block_summary.txt 2>&1
rm -rf $O/pmc_block_0 $O/pmc_block_1 $O/pmc_block_2 $O/pmc_block_3
tail -n 5 $O/retest.log; cat $O/skinny_probe.txt; cat $O/pmc_block_summary.txt
