#!/bin/bash
# r03 pass H: where a K = 1024 tile's time goes: one tile per CU (4096 x 4096 x 1024), instrumented and plain builds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python tools/gemm_timeline.py - 4096 4096 1024 > $O/h_timeline_one_tile.log 2>&1
timeout 300 python - > $O/h_one_tile_plain.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from midi_model_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(4096, 4096, 1024), (4096, 4096, 2048), (4096, 4096, 4096), (4096, 4096, 64), (8192, 4096, 1024), (16384, 4096, 1024)]:
    a = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn((N, K), device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    for abl in (0, 32):
        if abl == 32:
            ops.set_option("gemm_k64", 0)   # the no-store build exists for the K-step-32 loop only
        for k64 in ((1, 0) if abl == 0 else (0,)):
            ops.set_option("gemm_k64", k64)
            ops.set_option("gemm_ablate", abl)
            for _ in range(3):
                ops.gemm_nt(a, b, out, splitk=1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(a, b, out, splitk=1)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            print(f"M={M} N={N} K={K} k64={k64} ablate={abl}: {us:.1f} us per launch, {2.0 * M * N * K / us / 1e6:.0f} TF/s")
    ops.set_option("gemm_ablate", 0); ops.set_option("gemm_k64", 1)
PY
grep -v amdgpu $O/h_timeline_one_tile.log | tail -8; grep -v amdgpu $O/h_one_tile_plain.log
