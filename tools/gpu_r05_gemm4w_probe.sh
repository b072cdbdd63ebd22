#!/bin/bash
# tools/gemm4w_probe.hip against the production GEMM on the GPU box (build the probe first: see its header); profiles/r05_gemm4w_probe.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=gpurun_out
timeout 600 tools/bin/gemm4w_probe > $O/r05_probe_gemm4w.txt 2>&1; echo "probe rc=$?" >> $O/r05_probe_gemm4w.txt; cat $O/r05_probe_gemm4w.txt | cut -c1-330
