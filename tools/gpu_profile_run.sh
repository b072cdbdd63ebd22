#!/bin/bash
# One profile run for profiles/: GPU suite, default bench line (CPU baseline included), rocprofv3 kernel stats of the same
# command, PMC traffic passes, attention forms A/B.  Usage: tools/gpu_profile_run.sh <tag>   (outputs gpurun_out/<tag>_*)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
T=${1:-prof}
R=$(pwd)
timeout 2400 python -m pytest tests/ -q -m gpu --tb=short -p no:cacheprovider > $O/${T}_gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log
tail -n 3 $O/${T}_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${T}_smoke.log; tail -n 2 $O/${T}_smoke.log
timeout 1200 python bench.py > $O/${T}_bench.json 2> $O/${T}_gemm_by_shape.txt
echo "bench rc=$?"
(cd /tmp && rm -rf /tmp/prof_$T && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${T}_rocprof_bench.log 2>&1)
cp $(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1) $O/${T}_rocprofv3_kernel_stats.csv 2>/dev/null
bash tools/gpu_pmc_bench.sh > /dev/null 2>&1
cp $O/pmc_bench_traffic.txt $O/${T}_pmc_training_step_traffic_by_kernel.txt; cp $O/pmc_gemm_traffic.json $O/${T}_pmc_gemm_traffic.json
timeout 300 python tools/bench_attn_forms.py > $O/${T}_attn_forms_ab.txt 2>&1
# the north_star block line with its own rocprofv3 kernel stats; the other configurations of SURVEY.md 8(d); the vendor library
timeout 300 python bench.py --mode block > $O/${T}_block_bench.json 2>/dev/null
(cd /tmp && rm -rf /tmp/profb_$T && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb_$T -o block -- python $R/bench.py --mode block > /dev/null 2>&1)
cp $(find /tmp/profb_$T -name "*kernel_stats.csv" | head -1) $O/${T}_block_rocprofv3_kernel_stats.csv 2>/dev/null
# (BASELINE configs[4] -- tv2o-large and the 2x-hidden shape at 16 x 4096 per GPU -- are the `large` / `large_2x_hidden` objects of
#  the default bench line above since r04; the vendor-library comparison of r03 is profiles/r03_run3_hipblaslt_same_gpu.txt)
python - <<PY
import json
d=json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1]); a=d['attention']; b=d.get('block',{}); g=d.get('generate',{})
print(round(d['value']), 'ev/s', round(d['ms_per_step'],2), 'ms; gemm TF', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],4), '; attn fwd', round(a['fwd_us_per_layer'],1), 'bwd', round(a['bwd_us_per_layer'],1), 'us/layer; block', round(b.get('ms_per_block',0),3), 'ms frac', round(b.get('roofline',{}).get('frac',0),4), '; gen', round(g.get('value',0)), 'ev/s frac', round(g.get('roofline',{}).get('frac',0),4), '; cpu', d.get('cpu_baseline',{}).get('value'))
PY
head -12 $O/${T}_rocprofv3_kernel_stats.csv | cut -c1-150
