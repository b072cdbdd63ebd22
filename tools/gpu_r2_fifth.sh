#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
R=$(pwd)
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_decode_gpu.py "tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length" "tests/test_parity_long_gpu.py::test_projection_gemm_at_benchmarked_shapes" -q -m gpu --tb=short -p no:cacheprovider > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
# A/B of the GEMM transpose-read change on this box: r1 build of gemm_pp256 vs the current one, interleaved
for i in 1 2; do
  MH_LIB_PATH=$R/tools/bin/libmidihip_gemm_r1.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/train_old_$i.log 2> $O/train_old_$i.err
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/train_new_$i.log 2> $O/train_new_$i.err
done
timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline > $O/gen.log 2> $O/gen.err
(cd /tmp && rm -rf /tmp/gprof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gprof -o gen -- python $R/bench.py --mode generate --steps 1 --warmup 1 --gen-events 256 --no-cpu-baseline > $R/$O/gen_prof.log 2>&1)
python tools/trace_summary.py $(find /tmp/gprof -name "*kernel_trace.csv" | head -1) 0.4 > $O/gen_trace_summary.txt 2>&1
tail -n 6 $O/retest.log
python - <<PY
import json
for tag in ("old_1","new_1","old_2","new_2"):
    try:
        d=json.loads(open("$O/train_%s.log"%tag).read().strip().splitlines()[-1])
        print(tag, round(d["value"]), "ev/s", round(d["ms_per_step"],2), "ms  gemm TF", round(d["roofline"]["achieved"],1), "plain", round(d["roofline"]["achieved_plain_epilogue"],1))
    except Exception as e: print(tag, "failed", e)
d=json.loads(open("$O/gen.log").read().strip().splitlines()[-1]); print("generate", round(d["value"]), "ev/s", d["config"]["ms_per_event_step"], "ms/event frac", round(d["roofline"]["frac"],4))
PY
head -24 $O/gen_trace_summary.txt
