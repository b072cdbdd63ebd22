#!/bin/bash
# Round-2 first GPU pass: new parity tests, the full GPU suite, bench (train + block + generate objects), decode probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_parity_long_gpu.py tests/test_decode_gpu.py -q -m gpu --tb=short -p no:cacheprovider -s --timeout 600 > $O/new_tests.log 2>&1
echo "new tests rc=$?" >> $O/new_tests.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu --tb=short -p no:cacheprovider --timeout 600 -x > $O/old_tests.log 2>&1
echo "old tests rc=$?" >> $O/old_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
timeout 600 python tools/decode_probe.py > $O/decode_probe.txt 2>&1
echo "probe rc=$?" >> $O/decode_probe.txt
tail -n 4 $O/new_tests.log; tail -n 3 $O/old_tests.log; tail -c 1500 $O/bench.log; tail -n 30 $O/decode_probe.txt
