#!/bin/bash
# session 2, run 1: third form of the attention kernels -- parity tests, per-kernel A/B against the first form, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py::test_attention_fwd_bwd tests/test_kernels_gpu.py::test_attention_mfma_vs_plain_on_device tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length -q -m gpu --tb=short -p no:cacheprovider -x > $O/s2_1_attn_tests.log 2>&1
echo "attn tests rc=$?" >> $O/s2_1_attn_tests.log
tail -n 15 $O/s2_1_attn_tests.log
timeout 300 python tools/bench_attn_forms.py > $O/s2_1_attn_forms.txt 2>&1
cat $O/s2_1_attn_forms.txt
for v3 in 0 7; do
  MH_ATTN_V3=$v3 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>$O/s2_1_bench_v3_$v3.err > $O/s2_1_bench_v3_$v3.json
  python - <<PY
import json
d=json.loads(open("$O/s2_1_bench_v3_$v3.json").read().strip().splitlines()[-1]); a=d['attention']; b=d.get('block',{})
print('v3=$v3', round(d['value']), 'ev/s', round(d['ms_per_step'],2), 'ms; attn fwd', round(a['fwd_us_per_layer'],1), 'bwd', round(a['bwd_us_per_layer'],1), 'us/layer; block', round(b.get('ms_per_block',0),3), 'ms frac', round(b.get('roofline',{}).get('frac',0),4))
PY
done
