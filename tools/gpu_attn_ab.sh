#!/bin/bash
# event-level attention: parity tests of every kernel form, then the per-kernel A/B on this box (tools/bench_attn_forms.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py::test_attention_fwd_bwd tests/test_kernels_gpu.py::test_attention_forward_when_the_reference_has_to_move tests/test_kernels_gpu.py::test_attention_mfma_vs_plain_on_device tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length -q -m gpu --tb=short -p no:cacheprovider > $O/attn_ab_attn_tests.log 2>&1
echo "attn tests rc=$?" >> $O/attn_ab_attn_tests.log
tail -n 12 $O/attn_ab_attn_tests.log
MH_BENCH_ABLATE=0 timeout 300 python tools/bench_attn_forms.py 2>&1 | grep "S=" > $O/attn_ab_attn_forms.txt
cat $O/attn_ab_attn_forms.txt
