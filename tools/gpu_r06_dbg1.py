import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from midi_model_amd import ops
import emu_ops as emu
def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)
dtype = torch.bfloat16
for (M, N, K, ta, tb) in [(1024, 256, 2051, True, False), (1024, 256, 2048, True, False), (1024, 512, 2051, True, True), (512, 512, 1024, False, False)]:
    Kn = (K + 7) // 8 * 8
    a = rnd((K, M), dtype, 41) if ta else torch.cat([rnd((M, K), dtype, 41), torch.zeros((M, Kn - K), dtype=dtype)], 1)
    b = rnd((K, N), dtype, 42) if tb else torch.cat([rnd((N, K), dtype, 42), torch.zeros((N, Kn - K), dtype=dtype)], 1)
    want = emu.gemm_nt(a, b, torch.empty((M, N), dtype=dtype), K=K, ta=ta, tb=tb).float()
    for lean in (0, 1, 1, 1):
        for k64 in (1, 0):
            ops.set_option("gemm_lean_epi", lean)
            ops.set_option("gemm_k64", k64)
            out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
            ops.gemm_nt(a.cuda(), b.cuda(), out, K=K, ta=ta, tb=tb, splitk=1)
            err = (out.float().cpu() - want).abs()
            bad = (err > 0.5 + 0.05 * want.abs()) | ~torch.isfinite(out.float().cpu())
            idx = bad.nonzero()
            print(f"{M}x{N}x{K} ta={ta} tb={tb} lean={lean} k64={k64}: bad {int(bad.sum())}", idx[:12].tolist(), flush=True)
ops.set_option("gemm_lean_epi", 1); ops.set_option("gemm_k64", 1)
