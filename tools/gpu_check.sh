#!/bin/bash
# One GPU-box pass: hardware probe, per-kernel parity, end-to-end parity, smoke, bench (+ optional rocprof).
# Everything lands in gpurun_out/ (merged back by gpurun).  Usage: tools/gpu_check.sh [quick|full|prof]
MODE=${1:-full}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
tools/bin/probe > gpurun_out/probe.txt 2>&1
echo "probe rc=$?" >> gpurun_out/probe.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -n 1 --timeout 300 -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short -n 1 --timeout 600 -p no:cacheprovider -s > gpurun_out/model.log 2>&1
echo "model rc=$?" >> gpurun_out/model.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
if [ "$MODE" != "quick" ]; then
  timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench rc=$?" >> gpurun_out/bench.err
fi
if [ "$MODE" = "prof" ]; then
  cd /tmp
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof"
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.log" 2>&1
  echo "prof rc=$?" >> "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.log"
fi
for f in kernels model smoke; do tail -n 3 gpurun_out/$f.log; done
cat gpurun_out/bench.log 2>/dev/null | tail -2
