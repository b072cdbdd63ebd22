#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$(pwd)
timeout 1200 python -m pytest tests/test_kernels_gpu.py::test_attention_fwd_bwd tests/test_kernels_gpu.py::test_attention_mfma_vs_plain_on_device "tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length" tests/test_decode_gpu.py "tests/test_kernels_gpu.py::test_sample_top_p_k_fused" "tests/test_kernels_gpu.py::test_gemm_skinny" tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
for f in 1 2; do
  MH_ATTN_FWD=$f timeout 300 python bench.py --mode block --steps 10 --warmup 3 > $O/block_fwd$f.log 2> $O/block_fwd$f.err
done
(cd /tmp && rm -rf /tmp/gprof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gprof -o gen -- python $R/bench.py --mode generate --steps 1 --warmup 1 --gen-events 256 --no-cpu-baseline > $R/$O/gen_prof.log 2>&1)
cp $(find /tmp/gprof -name "*kernel_stats.csv" | head -1) $O/gen_kernel_stats.csv 2>/dev/null
python tools/trace_summary.py $(find /tmp/gprof -name "*kernel_trace.csv" | head -1) 0.4 > $O/gen_trace_summary.txt 2>&1
tail -n 8 $O/retest.log
for f in 1 2; do python - <<PY
import json
d=json.loads(open("$O/block_fwd$f.log").read().strip().splitlines()[-1])
b=d["block"]; print("MH_ATTN_FWD=$f", "ms", round(b["ms_per_block"],3), "frac", round(b["roofline"]["frac"],4), "attn TF", round(b["attention_tflops"],1), {k:round(v["us_per_call"],1) for k,v in b["kernels"].items()})
PY
done
head -25 $O/gen_trace_summary.txt; tail -2 $O/gen_prof.log | cut -c1-600
