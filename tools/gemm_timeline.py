#!/usr/bin/env python
"""In-kernel timeline of the K-step-64 main loops of gemm_pp256_kernel (ABL bit 7 builds): lane 0 of every wave of workgroup 0
stamps s_memtime at the seams of its load / MFMA slots; this prints the average shader cycles per segment, per phase of the
K-tile, for one wave of each ping-pong group.  usage: gemm_timeline.py - [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

_AB = ops.ab_library()  # a measurement tool: the compared kernel forms live in libmidihip_ab.so (build.py, -DMH_AB_BUILDS)
_AB.__enter__()
from midi_model_amd.lib import lib  # noqa: E402

variants = [1]
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (32768, 1024, 4096)
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
b = torch.randn((N, K), device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
LABELS0 = ["reads issued", "LDS-DMA issued", "lds wait", "barrier", "MFMA issued", "vmcnt wait", "barrier"]
LABELS1 = ["MFMA issued", "barrier", "reads issued", "LDS-DMA issued", "lds wait", "vmcnt wait", "barrier"]
LABELS_B = ["LDS-DMA issued", "reads issued"]  # OPT bit 1 builds (variants 4, 5) swap the first two of a load slot
PER = 7
for v in variants:
    ops.set_option("gemm_ablate", 128)
    ws = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
    for _ in range(2):
        lib().call("mh_gemm", a.data_ptr(), K, 0, b.data_ptr(), K, 0, out.data_ptr(), N, None, 0, M, N, K, 1.0, 0.0, 1, 1,
                   ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = ws.cpu().view(8, 256)
    ops.set_option("gemm_ablate", 0)
    print(f"== variant {v} M={M} N={N} K={K}: shader cycles per segment, averaged over phases 8.. (two phases per K-tile)")
    nphase = min(252 // PER - 1, 2 * ((K + 63) // 64) - 1)
    for w in (0, 4):
        names = list(LABELS0 if w < 4 else LABELS1)
        s = st[w].tolist()
        acc = [[0.0] * PER for _ in range(2)]
        cnt = [0, 0]
        for q in range(8, nphase - 1):
            for i in range(PER):
                acc[q % 2][i] += s[q * PER + i + 1] - s[q * PER + i]
            cnt[q % 2] += 1
        for p in range(2):
            row = ", ".join(f"{names[i]} {acc[p][i] / max(1, cnt[p]):.0f}" for i in range(PER))
            print(f"  wave {w} phase {p}: {row}  | total {sum(acc[p]) / max(1, cnt[p]):.0f}")
        print(f"  wave {w}: {(s[(nphase - 1) * PER] - s[8 * PER]) / (nphase - 1 - 8) * 2:.0f} cycles per K-tile")
    t0 = st[0, 16 * PER].item()
    for w in range(8):
        print(f"  wave {w} stamps, phases 16..17 on wave 0's clock:", [x - t0 for x in st[w, 16 * PER:18 * PER + 1].tolist()])
    # one tile per CU: where a tile's time goes (host-timed launch against the stamps of workgroup 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib().call("mh_gemm", a.data_ptr(), K, 0, b.data_ptr(), K, 0, out.data_ptr(), N, None, 0, M, N, K, 1.0, 0.0, 1, 1,
                   ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    for w in (0, 4):
        ent, land, done = (st[w, i].item() for i in (252, 253, 254))
        print(f"  wave {w}: entry -> first regions landed {land - ent} cycles, main loop {done - land} cycles ({tiles} tiles, "
              f"{e0.elapsed_time(e1) * 100:.1f} us per launch of the instrumented build)")
