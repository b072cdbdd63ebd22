#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python tools/decode_probe.py 1b > $O/decode_probe_1b.txt 2>&1
timeout 200 tools/bin/skinny_probe > $O/skinny_probe_plain.txt 2>&1
timeout 200 tools/bin/skinny_probe_preload > $O/skinny_probe_preload.txt 2>&1
timeout 600 python -m pytest "tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length" -q -m gpu --tb=line -p no:cacheprovider > $O/retest.log 2>&1
cat $O/decode_probe_1b.txt; grep "us per launch" $O/skinny_probe_plain.txt | head -14 | cut -c1-110; echo ---; grep "us per launch\|empty" $O/skinny_probe_preload.txt | head -19 | cut -c1-110; tail -3 $O/retest.log
