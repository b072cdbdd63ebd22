#!/bin/bash
# r03 pass G: the new parity tests at the benchmarked batch shapes + the default bench line with the K-step-64 loop.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_parity_long_gpu.py -q -m gpu --tb=short --timeout 900 -p no:cacheprovider -s -k "benchmarked_batch or ragged or tv2o_large" > $O/g_parity.log 2>&1
echo "parity rc=$?" >> $O/g_parity.log
timeout 900 python -m pytest tests/test_decode_gpu.py -q -m gpu --tb=short --timeout 600 -p no:cacheprovider -s -k "production_decode" > $O/g_decode.log 2>&1
echo "decode rc=$?" >> $O/g_decode.log
timeout 900 python bench.py > $O/g_bench.json 2> $O/g_bench.err
echo "bench rc=$?" >> $O/g_bench.err
tail -5 $O/g_parity.log; tail -5 $O/g_decode.log; cut -c1-600 $O/g_bench.json
