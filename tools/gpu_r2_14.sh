#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_kernels_gpu.py::test_attention_fwd_bwd tests/test_kernels_gpu.py::test_attention_mfma_vs_plain_on_device "tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length" "tests/test_parity_long_gpu.py::test_training_step_gradients_at_S2048" -q -m gpu --tb=short -p no:cacheprovider > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
for i in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MH_LIB_PATH=$R/tools/bin/libmidihip_attn_prev.so; else unset MH_LIB_PATH; fi
    timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['attention']; print('$lib', round(d['value']), 'ev/s', round(d['ms_per_step'],2), 'ms; attn fwd', round(a['fwd_us_per_layer'],1), 'bwd', round(a['bwd_us_per_layer'],1), 'us/layer')"
  done
done
unset MH_LIB_PATH
tail -n 4 $O/retest.log
