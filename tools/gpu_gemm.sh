#!/bin/bash
# GPU-box pass focused on the projection GEMM: kernel parity tests, per-shape micro-benchmark of both kernel
# variants, then the end-to-end bench with each variant.  Output in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -n 1 --timeout 300 -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log
timeout 900 python tools/bench_gemm.py ${1:-0,1} ${2:-0} > gpurun_out/bench_gemm.log 2>&1
echo "bench_gemm rc=$?" >> gpurun_out/bench_gemm.log
for v in 0 1; do
  MH_GEMM=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_v$v.log 2> gpurun_out/bench_v$v.err
  echo "bench v$v rc=$?" >> gpurun_out/bench_v$v.err
done
tail -4 gpurun_out/kernels.log
tail -3 gpurun_out/bench_gemm.log
cut -c1-400 gpurun_out/bench_v0.log gpurun_out/bench_v1.log
