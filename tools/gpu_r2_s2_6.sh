#!/bin/bash
# session 2, run 6: delta folded into the dP MFMA chains (dQ, dK/dV), packed conversions -- parity tests, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py::test_attention_fwd_bwd tests/test_kernels_gpu.py::test_attention_forward_when_the_reference_has_to_move tests/test_kernels_gpu.py::test_attention_mfma_vs_plain_on_device tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length -q -m gpu --tb=short -p no:cacheprovider > $O/s2_6_attn_tests.log 2>&1
echo "attn tests rc=$?" >> $O/s2_6_attn_tests.log
tail -n 12 $O/s2_6_attn_tests.log
MH_BENCH_ABLATE=0 timeout 300 python tools/bench_attn_forms.py 2>&1 | grep "S=" > $O/s2_6_attn_forms.txt
cat $O/s2_6_attn_forms.txt
