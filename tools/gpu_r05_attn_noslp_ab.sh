#!/bin/bash
# attention kernels built with -fno-slp-vectorize (no compiler-formed v_pk_*_f32 beside the MFMAs) against the tree's build, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=gpurun_out
for i in 1 2 3; do
  python tools/attn_fwd_once.py >> $O/r05_attn_noslp_ab.txt 2>/dev/null
  MH_LIB_PATH=$(pwd)/tools/bin/libmidihip_noslp.so python tools/attn_fwd_once.py >> $O/r05_attn_noslp_ab.txt 2>/dev/null
done
cat $O/r05_attn_noslp_ab.txt
