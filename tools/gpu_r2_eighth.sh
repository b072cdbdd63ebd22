#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$(pwd)
timeout 900 python -m pytest "tests/test_kernels_gpu.py::test_sample_top_p_k_fused" "tests/test_kernels_gpu.py::test_gemm_skinny" tests/test_kernels_gpu.py::test_masked_softmax tests/test_decode_gpu.py tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline > $O/gen.log 2> $O/gen.err
(cd /tmp && rm -rf /tmp/gprof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gprof -o gen -- python $R/bench.py --mode generate --steps 1 --warmup 1 --gen-events 256 --no-cpu-baseline > $R/$O/gen_prof.log 2>&1)
python tools/trace_summary.py $(find /tmp/gprof -name "*kernel_trace.csv" | head -1) 0.4 > $O/gen_trace_summary.txt 2>&1
tail -n 5 $O/retest.log
python - <<PY
import json
d=json.loads(open("$O/gen.log").read().strip().splitlines()[-1]); print("generate", round(d["value"]), "ev/s", d["config"]["ms_per_event_step"], "ms/event frac", round(d["roofline"]["frac"],4))
PY
head -24 $O/gen_trace_summary.txt
