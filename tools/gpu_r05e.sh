#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu --tb=short -p no:cacheprovider > $O/r05e_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/r05e_gpu_tests.log; tail -15 $O/r05e_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
