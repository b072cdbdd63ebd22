#!/bin/bash
# full GPU suite + the driver-shaped bench + rocprofv3 stats + PMC traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$(pwd)
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/full_tests.log 2>&1
echo "full tests rc=$?" >> $O/full_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
(cd /tmp && rm -rf /tmp/tprof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tprof -o train -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/train_prof.log 2>&1)
cp $(find /tmp/tprof -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv 2>/dev/null
python tools/trace_summary.py $(find /tmp/tprof -name "*kernel_trace.csv" | head -1) 0.5 > $O/train_trace_summary.txt 2>&1
tools/gpu_pmc_bench.sh > $O/pmc_bench.log 2>&1
tail -n 4 $O/full_tests.log; tail -n 2 $O/smoke.log; tail -c 600 $O/bench.log; echo; cat $O/pmc_gemm_traffic.json; head -12 $O/train_trace_summary.txt
