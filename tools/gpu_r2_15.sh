#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline > $O/gen.log 2> $O/gen.err
echo "gen rc=$?" >> $O/gen.err
timeout 300 python tools/decode_probe.py 1b 2 > $O/decode_probe15.txt 2>&1
tail -n 12 $O/retest.log; tail -c 1500 $O/gen.log; tail -3 $O/gen.err; cat $O/decode_probe15.txt
