#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=gpurun_out
timeout 600 tools/bin/gemm4w_probe > $O/r05c_gemm4w.txt 2>&1; echo "probe rc=$?" >> $O/r05c_gemm4w.txt; cat $O/r05c_gemm4w.txt | cut -c1-330
