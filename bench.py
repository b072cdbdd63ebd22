#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: MIDI events/sec of the tv2o-medium training step (BASELINE.json
configs[1]: bf16, per-GPU batch 16, 2048 events per sequence), one process per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + backward + gradient all-reduce (N>1) + global-norm clip + AdamW on one synthetic batch
(random-init weights; no dataset/checkpoint is reachable).  Rank 0 prints ONE JSON line.  Extra objects:
  roofline      the projection GEMM kernel (the dominant kernel): algorithmic FLOPs / HIP-event time, vs the
                2.5 PFLOP/s dense bf16 MFMA peak;
  cpu_baseline  the CPU oracle (fp32 torch restatement of the reference step) timed on this box's host cores
                on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def train_flops_per_event(S: int, net_L=12, tok_L=3, D=1024, I=4096, It=1024, V=3406) -> float:
    """fwd FLOPs/event (2*MAC, causal attention on the lower triangle) x3 for fwd+bwd — BASELINE.md §3."""
    net_proj = net_L * 2 * (4 * D * D + 3 * D * I)
    net_attn = net_L * 4 * D * (S + 1) / 2
    tok_proj = 8 * tok_L * 2 * (4 * D * D + 3 * D * It)
    tok_attn = tok_L * 2 * 2 * D * 36  # 36 causal (query,key) pairs per octet, QK^T and PV, 2 FLOP/MAC
    lm = 8 * 2 * D * V
    return 3.0 * (net_proj + net_attn + tok_proj + tok_attn + lm)


def cpu_baseline(sample_S: int, seed: int = 0):
    """The oracle's training step (fwd + autograd bwd + clip + AdamW, fp32) on the host cores, B=1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("midi_oracle", os.path.join(ROOT, "oracle", "midi_oracle.py"))
    orc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(orc)
    import midi_model_amd as mm
    # With one thread per core of the GPU box's 256-core host the step collapsed to 2.4 events/s (421 s for
    # 1024 events: oversubscribed GEMMs, profiles/r01_run1_bench.json); the baseline is therefore bounded to
    # 32 threads and a shorter sample.  `cores` reports the threads actually used.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    tok = mm.MIDITokenizerV2()
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = {k: v.requires_grad_(True) for k, v in orc.make_state_dict(shp, seed=seed).items()}
    batch = orc.synthetic_events(tok, 1, sample_S + 1, seed=seed)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    t0 = time.perf_counter()
    loss, _ = orc.training_loss(sd, shp, batch)
    loss.backward()
    coef, _ = orc.clip_coef([p.grad for p in sd.values()], 1.0)
    with torch.no_grad():
        for k, p in sd.items():
            orc.adamw_step(p, p.grad * coef, m[k], v2[k], 1, 2e-4, 0.01 if orc.decays(k) else 0.0)
    dt = time.perf_counter() - t0
    return {"value": sample_S / dt, "unit": "events/s", "cores": cores, "kind": "port",
            "sample": f"1 training step (fwd+bwd+clip+AdamW) of the CPU oracle, fp32, batch 1 x {sample_S} events, {dt:.1f} s",
            "loss": float(loss.detach())}


def cpu_baseline_generate(batch: int, n_events: int, seed: int = 0):
    """The oracle's KV-cached generate() (fp32, host cores): `batch` sequences x `n_events` new events, EOS masked."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("midi_oracle", os.path.join(ROOT, "oracle", "midi_oracle.py"))
    orc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(orc)
    import midi_model_amd as mm
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    tok = mm.MIDITokenizerV2()
    shp = orc.Shape(vocab=tok.vocab_size)
    with torch.no_grad():
        sd = orc.make_state_dict(shp, seed=seed)
        t0 = time.perf_counter()
        out = orc.generate(sd, shp, tok, None, batch_size=batch, max_len=1 + n_events, generator=torch.Generator().manual_seed(seed),
                           ban_eos=True)
        dt = time.perf_counter() - t0
    assert out.shape[1] == 1 + n_events
    return {"value": batch * n_events / dt, "unit": "events/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle generate(), fp32, batch {batch} x {n_events} new events, {dt:.1f} s"}


def decode_bytes_per_event(B: int, n_cached: float, token_steps: float, L=12, D=1024, I=4096, Lt=3, It=1024, V=3406) -> float:
    """SURVEY.md 8(d) algorithmic bytes of one generated event at batch B with n cached events, bf16: the net's
    non-embedding weights once, its K/V cache once, and net_token + lm_head weights once per token step."""
    w_net = 2.0 * L * (4 * D * D + 3 * D * I)
    kv = 2.0 * 2 * L * B * n_cached * D
    w_tok = 2.0 * (Lt * (4 * D * D + 3 * D * It) + V * D)
    return w_net + kv + token_steps * w_tok


def bench_generate(args):
    """BASELINE.json configs[3]: generate(), batch 64, BOS prompt, 1024 new events, temp 1 / top_p 0.98 / top_k 20,
    seeded, EOS masked so that every row runs the full length (ban_eos; stated in config).  One step = one generate()
    call.  Replicas only across GPUs: every rank generates its own batch, no collective."""
    import midi_model_amd as mm
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU implementation)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = mm.MIDIModel(mm.MIDIModelConfig.from_name(args.config)).to(torch.device("cuda", local), dtype).eval()
    B, n_new = args.gen_batch, args.gen_events
    gen = torch.Generator(device="cuda")

    def run(n_events, seed):
        gen.manual_seed(seed)
        with torch.no_grad():
            return model.generate(None, batch_size=B, max_len=1 + n_events, temp=1.0, top_p=0.98, top_k=20, generator=gen,
                                  ban_eos=True)

    for i in range(args.warmup):  # full length: the decode session (K/V capacity, captured graphs) is sized by max_len
        run(n_new, 100 + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tok_steps = 0
    for i in range(args.steps):
        out = run(n_new, 1000 * rank + i)
        assert out.shape == (B, 1 + n_new, 8), out.shape
        tok_steps += int((out[:, 1:, :] != 0).any(axis=0).sum())  # token positions some row filled (lower bound of steps run)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        events = world * B * n_new * args.steps
        per_event_s = dt / (n_new * args.steps)
        steps_per_event = tok_steps / (n_new * args.steps)
        by = decode_bytes_per_event(B, (1 + n_new) / 2.0, steps_per_event)
        out_d = {
            "metric": f"MIDI events/sec, KV-cached generate(), {args.config}", "value": events / dt, "unit": "events/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.config} {args.dtype} generate(): batch {B} per GPU, BOS prompt, {n_new} new events, "
                                   f"temp 1.0 top_p 0.98 top_k 20, seeded; EOS masked so every row runs the full length "
                                   f"(BASELINE.json configs[3]); random-init weights",
                       "global_batch": world * B, "new_events": n_new, "parallelism": f"replicas x{world}",
                       "ms_per_event_step": 1e3 * per_event_s, "token_steps_per_event": steps_per_event},
            "roofline": {"bound": "hbm", "kernel": "decode step (weights + K/V cache streamed once per event / token step)",
                         "achieved": by / per_event_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": by / per_event_s / 8.0e12, "traffic": None, "algorithmic_bytes_per_event_step": by},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out_d["cpu_baseline"] = cpu_baseline_generate(B, 32)  # B=64 for 32 events
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out_d["cpu_baseline"] = {"value": None, "unit": "events/s", "cores": os.cpu_count(), "kind": "port",
                                         "sample": f"failed: {e!r}"}
        print(json.dumps(out_d))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (sequences)")
    ap.add_argument("--seq", type=int, default=2048, help="events per sequence seen by the model")
    ap.add_argument("--config", default="tv2o-medium")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-sample-seq", type=int, default=2048,
                    help="events in the CPU baseline sample (one sequence of the workload: ~10-20 s of host time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "generate"],
                    help="train: the headline metric (BASELINE.json configs[1]); generate: KV-cached generate(), configs[3]")
    ap.add_argument("--gen-batch", type=int, default=64)
    ap.add_argument("--gen-events", type=int, default=1024, help="new events per sequence per generate() call")
    args = ap.parse_args()
    if args.mode == "generate":
        return bench_generate(args)

    import midi_model_amd as mm
    from midi_model_amd import ops
    from midi_model_amd.data import synthetic_events
    from midi_model_amd.train import TrainMIDIModel
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU implementation)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    cfg = mm.MIDIModelConfig.from_name(args.config)
    model = TrainMIDIModel(cfg, lr=2e-4, weight_decay=0.01, warmup=1e3, max_step=1e6, accumulate_grad_batches=1)
    model = model.to(torch.device("cuda", local), dtype)
    model.configure_optimizers()
    model.broadcast_parameters(0)
    B, S = args.batch, args.seq
    batches = [synthetic_events(model.tokenizer, B, S + 1, seed=1000 + 17 * rank + i, device="cuda") for i in range(2)]

    def step(i):
        loss = model.training_step(batches[i % 2])
        model.optimizer_step()
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prof = None if args.no_gemm_events else []
    ops.gemm_profile = prof
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.gemm_profile = None
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_v = float(loss.item())

    if rank == 0:
        events = world * B * S * args.steps
        value = events / dt
        nc, tc = cfg.net_config, cfg.net_token_config
        fl_event = train_flops_per_event(S, net_L=nc.num_hidden_layers, tok_L=tc.num_hidden_layers, D=nc.hidden_size,
                                         I=nc.intermediate_size, It=tc.intermediate_size, V=model.tokenizer.vocab_size)
        out = {
            "metric": f"MIDI events/sec, training step (fwd+bwd+clip+AdamW), {args.config}, seq={S}",
            "value": value, "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.config} {args.dtype} training step, per-GPU batch {B} x {S} events x 8 tokens "
                                   f"({'BASELINE.json configs[1]' if (args.config, B, S) == ('tv2o-medium', 16, 2048) else 'non-headline configuration'}); "
                                   f"random-init weights, synthetic events",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}", "accumulate_grad_batches": 1,
                       "optimizer": "AdamW bf16-true + global-norm clip 1.0" if args.dtype == "bf16" else "AdamW fp32 + clip"},
            "loss": loss_v,
            "model_tflops_per_gpu": fl_event * B * S * args.steps / dt / 1e12,
            "model_flops_frac_of_peak": fl_event * B * S * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS,
        }
        if prof:
            ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in prof)
            fl = sum(f for _, _, f, _ in prof)
            n = len(prof)
            ach = fl / (ms * 1e-3) / 1e12
            by_shape = {}
            for e0, e1, f, shp in prof:
                t_, f_, n_ = by_shape.get(shp, (0.0, 0.0, 0))
                by_shape[shp] = (t_ + e0.elapsed_time(e1), f_ + f, n_ + 1)
            plain = [(e0.elapsed_time(e1), f) for e0, e1, f, shp in prof if len(shp) == 6]  # no elementwise work in the epilogue
            print("[bench] GEMM launches by shape (M,N,K,splitk,transA,transB[,fused epilogue]): calls, total ms, TFLOP/s", file=sys.stderr)
            for shp, (t_, f_, n_) in sorted(by_shape.items(), key=lambda kv: -kv[1][0]):
                print(f"[bench]   {shp}: {n_:4d} {t_:9.3f} {f_ / (t_ * 1e-3) / 1e12:8.1f}", file=sys.stderr)
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_pp256_kernel (all projection GEMMs: fwd, dgrad, wgrad, lm_head)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                               "traffic": None, "launches": n, "avg_launch_us": 1e3 * ms / n,
                               "avg_flops_per_launch": fl / n, "gemm_share_of_step_time": ms * 1e-3 / dt,
                               # the same ratio over the launches whose epilogue carries no SwiGLU forward / backward
                               "achieved_plain_epilogue": (sum(f for _, f in plain) / (sum(t for t, _ in plain) * 1e-3) / 1e12
                                                           if plain else None),
                               "launches_plain_epilogue": len(plain)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_seq)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "events/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
