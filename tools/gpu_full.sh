#!/bin/bash
# the whole GPU suite + the default bench line (no CPU baseline); logs under gpurun_out/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T=${1:-full}
timeout 2400 python -m pytest tests/ -q -m gpu --tb=short -p no:cacheprovider -x > $O/${T}_gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log
tail -n 6 $O/${T}_gpu_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>$O/${T}_bench.err > $O/${T}_bench.json
python - <<PY
import json
d=json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1]); a=d['attention']; b=d.get('block',{}); g=d.get('generate',{})
print(round(d['value']), 'ev/s', round(d['ms_per_step'],2), 'ms; gemm TF', round(d['roofline']['achieved'],1), '; attn fwd', round(a['fwd_us_per_layer'],1), 'bwd', round(a['bwd_us_per_layer'],1), 'us/layer; block', round(b.get('ms_per_block',0),3), 'ms frac', round(b.get('roofline',{}).get('frac',0),4), '; gen', round(g.get('value',0)), 'ev/s')
print({k: round(v['ms_per_step'],2) for k,v in d['kernel_families'].items()})
print({k: round(v['us_per_call'],1) for k,v in b.get('kernels',{}).items()})
PY
grep -A12 "by shape" $O/${T}_bench.err | head -30
