#!/usr/bin/env python
"""Where a generated event's time goes, and what the candidate structures buy (run on the GPU box; prints a report).

  1. per-launch cost of the decode projections inside a replayed hipGraph, for the three row tilings of mh_gemm_skinny
     ("skinny_mb" = 16-row activation blocks per workgroup: 4 = every workgroup takes all 64 rows, 1 = four times the
     workgroups with a quarter of the rows each), next to the floor of a graph node (a trivial kernel);
  2. the replay time of the session's graphs: net step at several cache lengths, token steps 0..7;
  3. aggregate throughput of G concurrent decode chains of 64/G sequences (threads x streams) -- what splitting the batch
     into independent chains buys when a chain is bound by dependent-launch latency.
"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_model_amd as mm  # noqa: E402
from midi_model_amd import ops  # noqa: E402
from midi_model_amd.decode import DecodeSession  # noqa: E402

dev = "cuda"
bf = torch.bfloat16


def graph_time(body, reps=20):
    """capture body() once, replay `reps` times, return ms per replay"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def section_1():
    print("== 1. decode projections inside a replayed graph (us per launch; 12 'layers' x 4 launches chained) ==")
    M, D = 64, 1024
    tiny = torch.zeros(64, device=dev)

    def floor_body():
        for _ in range(48):
            tiny.add_(1.0)
    print(f"graph node floor (trivial elementwise kernel): {1e3 * graph_time(floor_body) / 48:.2f} us per node")
    for I, what in ((4096, "event-level net (I=4096)"), (1024, "token-level net (I=1024)")):
        g = torch.Generator().manual_seed(0)
        wqkv = [(0.02 * torch.randn((3 * D, D), generator=g)).to(dev, bf) for _ in range(12)]
        wo = [(0.02 * torch.randn((D, D), generator=g)).to(dev, bf) for _ in range(12)]
        wgu = [(0.02 * torch.randn((2 * I, D), generator=g)).to(dev, bf) for _ in range(12)]
        wd = [(0.02 * torch.randn((D, I), generator=g)).to(dev, bf) for _ in range(12)]
        for rows in (64, 32, 16):
            x0 = torch.randn((rows, D), generator=g).to(dev, bf)
            qkv = torch.empty((rows, 3 * D), device=dev, dtype=bf)
            x2 = torch.empty((rows, D), device=dev, dtype=bf)
            a = torch.empty((rows, I), device=dev, dtype=bf)
            x3 = torch.empty((rows, D), device=dev, dtype=bf)

            def body():
                x = x0
                for l in range(12):
                    ops.gemm_skinny(x, wqkv[l], qkv, norm_eps=1e-6)
                    ops.gemm_skinny(qkv[:, :D], wo[l], x2, res=x)
                    ops.gemm_skinny(x2, wgu[l], a, mode=ops.SKINNY_GATEUP, norm_eps=1e-6)
                    ops.gemm_skinny(a, wd[l], x3, res=x2)
                    x = x3
            res = []
            for mb in (4, 2, 1):
                if mb * 16 > rows and mb != 1:
                    res.append("   -  ")
                    continue
                ops.set_option("skinny_mb", mb)
                res.append(f"{1e3 * graph_time(body) / 48:6.2f}")
            ops.set_option("skinny_mb", 0)
            wbytes = 2.0 * (4 * D * D + 3 * D * I)
            print(f"{what}, {rows} rows: mb=4 {res[0]}  mb=2 {res[1]}  mb=1 {res[2]} us/launch  "
                  f"(weights {wbytes / 1e6:.1f} MB per layer = {wbytes / 6.3e12 * 1e6:.1f} us at 6.3 TB/s for the 4 launches)")


def section_1b():
    print("== 1b. each decode projection alone, 48 chained launches in a replayed graph (us per launch), by row tiling ==")
    M, D = 64, 1024
    g = torch.Generator().manual_seed(0)
    shapes = [("q|k|v      N=3072 K=1024 rstd", 3072, 1024, 0, True, False),
              ("o / down   N=1024 K=1024 +res", 1024, 1024, 0, False, True),
              ("down (net) N=1024 K=4096 +res", 1024, 4096, 0, False, True),
              ("gate|up    I=4096 K=1024 rstd", 4096, 1024, 1, True, False),
              ("gate|up    I=1024 K=1024 rstd", 1024, 1024, 1, True, False),
              ("lm_head    N=3406 K=1024 rstd", 3406, 1024, 0, True, False)]
    for name, N, K, mode, rstd, res in shapes:
        ws = [(0.02 * torch.randn(((2 if mode else 1) * N, K), generator=g)).to(dev, bf) for _ in range(6)]
        x = torch.randn((M, K), generator=g).to(dev, bf)
        outs = [torch.zeros((M, (max(N, K) + 127) // 64 * 64), device=dev, dtype=bf) for _ in range(2)]
        r = torch.randn((M, N), generator=g).to(dev, bf) if res else None

        def body():
            a = x
            for l in range(48):
                o = outs[l & 1][:, :N]
                ops.gemm_skinny(a, ws[l % 6], o, mode=mode, norm_eps=1e-6 if rstd else 0.0, res=r)
                a = outs[l & 1][:, :K] if K <= N else x  # (chained where the shapes allow; K > N re-reads x)
        line = []
        for nbt in ((1, 2) if mode == 0 else (0,)):
            ops.set_option("skinny_nbt", nbt)
            for mb in (4, 2, 1):
                ops.set_option("skinny_mb", mb)
                line.append(f"nbt{nbt}/mb{mb} {1e3 * graph_time(body) / 48:5.2f}")
        ops.set_option("skinny_mb", 0)
        ops.set_option("skinny_nbt", 0)
        wb = 2.0 * (2 if mode else 1) * N * K
        print(f"{name}: " + "  ".join(line) + f"   ({wb / 1e6:5.1f} MB of weights = {wb / 6.3e12 * 1e6:.2f} us at 6.3 TB/s)")


def section_2(model):
    print("== 2. replay time of the session graphs (B=64, capacity 2048) ==")
    for mb in (4, 1):
        ops.set_option("skinny_mb", mb)
        with torch.inference_mode():
            ses = DecodeSession(model, 64, 2048, 1.0, 0.98, 20)
            ses.first_mask.copy_(model._grammar()[0])
            ses.first_mask[model.tokenizer.eos_id] = 0
            ses.reset()
            ses.begin(torch.Generator(device=dev).manual_seed(0))

            def t(fn, reps=10):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return 1e3 * e0.elapsed_time(e1) / reps
            toks = [t(lambda i=i: ses.tok_step(i)) for i in range(8)]
            nets = {}
            for pos in (1, 256, 1024, 2000):
                ses.pos.fill_(pos)
                nets[pos] = t(lambda: (ses.g_net.replay(), ses.pos.fill_(pos)))
            ses.end()
        print(f"skinny_mb={mb}: token steps (us) " + " ".join(f"{x:.0f}" for x in toks) + f"  sum {sum(toks):.0f};  net step (us) at cached "
              + ", ".join(f"{p}: {v:.0f}" for p, v in nets.items()))
        del ses
    ops.set_option("skinny_mb", 0)


def section_3(model, n_events=192):
    print(f"== 3. G concurrent chains of 64/G sequences, {n_events} new events each (events/s aggregate) ==")

    def run(B, seed, stream):
        with torch.cuda.stream(stream):
            g = torch.Generator(device=dev).manual_seed(seed)
            model.generate(None, batch_size=B, max_len=1 + n_events, generator=g, ban_eos=True)
            torch.cuda.current_stream().synchronize()
    for mb in (4, 1):
        ops.set_option("skinny_mb", mb)
        model._sessions.idle.clear()
        line = []
        for G in (1, 2, 4, 8):
            B = 64 // G
            streams = [torch.cuda.Stream() for _ in range(G)]
            for phase in ("warm", "timed"):
                ths = [threading.Thread(target=run, args=(B, 10 + i, streams[i])) for i in range(G)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                [th.start() for th in ths]
                [th.join() for th in ths]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            line.append(f"G={G} (B={B}): {64 * n_events / dt / 1e3:.1f}k")
            model._sessions.idle.clear()
        print(f"skinny_mb={mb}: " + "   ".join(line))
    ops.set_option("skinny_mb", 0)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["1", "2", "3"]
    if "1" in which:
        section_1()
    if "1b" in which:
        section_1b()
    model = mm.MIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium")).to(dev, bf).eval()
    if "2" in which:
        section_2(model)
    if "3" in which:
        section_3(model)
