#!/bin/bash
# r06 first GPU batch: kernel tests of the rewritten epilogues / lazy attention, same-box A/B against the r05 library, attention
# forward forms + in-kernel timeline, the block line, the long-shape parity tests the changes touch.
# (libmidihip_r05.so = the library built from the csrc/ of the last round-5 commit, `git worktree add /tmp/r05 7093e65` + build.py:
#  a git-ignored build product that is not kept in the tree; without it the A/B loop below only times this build.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06_b1}
python tools/gpu_r06_dbg1.py 2>&1 | grep -v amdgpu.ids | grep -v "bad 0" | head -20
timeout 1800 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -25 > gpurun_out/${T}_kernel_tests.txt
for i in 1 2 3; do
  [ -f midi-model_amd/libmidihip_r05.so ] && MH_LIB_PATH=$PWD/midi-model_amd/libmidihip_r05.so python tools/gemm_lib_once.py
  python tools/gemm_lib_once.py
done > gpurun_out/${T}_gemm_ab.txt 2>&1
python tools/attn_fwd_ab.py 127 255 > gpurun_out/${T}_attn_ab.txt 2>&1
(python tools/attn_timeline.py 4096 0; python tools/attn_timeline.py 4096 1; python tools/attn_timeline.py 2048 0) > gpurun_out/${T}_attn_timeline.txt 2>&1
python bench.py --mode block > gpurun_out/${T}_block.json 2> gpurun_out/${T}_block.err
timeout 2400 python -m pytest tests/test_parity_long_gpu.py -x -q -s -k "folded or swiglu_epilogues or flash_attention or public_forward" 2>&1 | tail -30 > gpurun_out/${T}_long_tests.txt
tail -n 5 gpurun_out/${T}_kernel_tests.txt; tail -n 5 gpurun_out/${T}_long_tests.txt; cat gpurun_out/${T}_attn_ab.txt | tail -12; tail -3 gpurun_out/${T}_gemm_ab.txt | cut -c1-1500
