#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: MIDI events/sec of the tv2o-medium training step (BASELINE.json
configs[1]: bf16, per-GPU batch 16, 2048 events per sequence), one process per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8                      # spawns 8 ranks itself (re-exec under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W    # (what the driver does; same thing)

A step = forward + backward + gradient all-reduce (N>1) + global-norm clip + AdamW on one synthetic batch
(random-init weights; no dataset/checkpoint is reachable).  Rank 0 prints ONE JSON line.  Objects in it:
  roofline         the projection GEMM kernel (the dominant kernel): algorithmic FLOPs / HIP-event time (split-K
                   reductions included), vs the 2.5 PFLOP/s dense bf16 MFMA peak; `traffic` = memory-side bytes per launch
                   from the committed rocprofv3 PMC pass (profiles/*_pmc_gemm_traffic.json, provenance in `traffic_source`)
  kernel_families  share of the step per kernel family, from HIP events around every C-ABI launch in the timed region
  attention        event-level flash attention forward / backward TFLOP/s from the same events
  allreduce_*      N>1: gradient bytes exchanged per step and the all-reduce time NOT hidden behind backward
  cpu_baseline     the CPU oracle (fp32 torch restatement of the reference step) timed on this box's host cores
                   on a bounded sample of the same workload (N=1, rank 0 only)
  block            N=1: the `north_star` target -- ONE net block forward (RMSNorm -> q|k|v -> RoPE -> flash attention ->
                   o + residual -> RMSNorm -> gate|up + SwiGLU -> down + residual) at batch 16 x 4096 events, fraction
                   of the bf16 MFMA peak against 41,945,088 FLOP per event; the forward-only form a
                   prefill of this size runs: the two RMSNorms folded around the projections (`norms_folded`;
                   `--mode block --block-unfolded` times the form with the two norm passes)
  large, large_2x_hidden   N=1: BASELINE.json configs[4] at its per-GPU batch (16 x 4096), with hbm_peak_gb / hbm_headroom_gb
  ranks, scaling_efficiency, comm_ab   N>1: every rank's own time, efficiency against --baseline-1gpu, --comm both
  generate         N=1: BASELINE.json configs[3] (KV-cached generate(), batch 64 x 1024 new events), fraction of the HBM
                   roofline of the decode step
`--mode block` / `--mode generate` print those measurements as the top-level line instead.
"""
import argparse
import gc
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0      # HBM3E spec (6.29 TB/s measured by a float4 copy, same guide)


def train_flops_per_event(S: int, net_L=12, tok_L=3, D=1024, I=4096, It=1024, V=3406) -> float:
    """fwd FLOPs/event (2*MAC, causal attention on the lower triangle) x3 for fwd+bwd — BASELINE.md §3."""
    net_proj = net_L * 2 * (4 * D * D + 3 * D * I)
    net_attn = net_L * 4 * D * (S + 1) / 2
    tok_proj = 8 * tok_L * 2 * (4 * D * D + 3 * D * It)
    tok_attn = tok_L * 2 * 2 * D * 36  # 36 causal (query,key) pairs per octet, QK^T and PV, 2 FLOP/MAC
    lm = 8 * 2 * D * V
    return 3.0 * (net_proj + net_attn + tok_proj + tok_attn + lm)


def block_flops_per_event(S: int, D=1024, I=4096) -> float:
    """one net block forward: projections 2*(4 D^2 + 3 D I) + causal attention 2 * 2 * D * (S+1)/2 (SURVEY.md 8(d):
    33,554,432 + 2048 (S+1); S = 4096 -> 41,945,088)"""
    return 2.0 * (4 * D * D + 3 * D * I) + 2.0 * D * (S + 1)


def _load_oracle():
    import importlib.util
    spec = importlib.util.spec_from_file_location("midi_oracle", os.path.join(ROOT, "oracle", "midi_oracle.py"))
    orc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(orc)
    return orc


def cpu_baseline(sample_S: int, seed: int = 0):
    """The oracle's training step (fwd + autograd bwd + clip + AdamW, fp32) on the host cores, B=1."""
    orc = _load_oracle()
    # attention through torch's fused scaled_dot_product_attention, the call the reference makes (TF:integrations/sdpa_attention.py):
    # with the oracle's explicit scores / softmax the timed step was 2x slower than the REAL reference on the same cores, with it
    # the two agree (tools/cpu_reference_vs_port.py ran both in the build container: profiles/r04_cpu_reference_vs_port.txt)
    orc.FUSED_SDPA = True
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
            "sample": f"1 training step (fwd+bwd+clip+AdamW) of the CPU oracle (attention through torch SDPA, as the reference), fp32, "
                      f"batch 1 x {sample_S} events, {dt:.1f} s; in the build container the REAL reference and this form agree within run-to-run "
                      f"noise (214-255 vs 214-247 events/s on 8 cores, profiles/r04_cpu_reference_vs_port.txt)",
            "loss": float(loss.detach())}


def cpu_baseline_config0(seed: int = 0):
    """BASELINE.json configs[0]: tv2o-medium MIDIModel.forward on the CPU, batch 1 x 128 events, fp32 (the reference's own
    CPU-runnable case) -- the oracle's forward on the host cores, best of three calls."""
    orc = _load_oracle()
    orc.FUSED_SDPA = True
    import midi_model_amd as mm
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    tok = mm.MIDITokenizerV2()
    shp = orc.Shape(vocab=tok.vocab_size)
    with torch.no_grad():
        sd = orc.make_state_dict(shp, seed=seed)
        x = orc.synthetic_events(tok, 1, 128, seed=seed)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            h = orc.midi_forward(sd, shp, x)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    assert tuple(h.shape) == (1, 128, shp.n_embd)
    return {"value": 128 / best, "unit": "events/s", "ms_per_call": 1e3 * best, "cores": cores, "kind": "port",
            "sample": "MIDIModel.forward, batch 1 x 128 events, fp32, CPU oracle (BASELINE.json configs[0]); best of 3 calls"}


def cpu_baseline_generate(batch: int, n_events: int, seed: int = 0):
    """The oracle's KV-cached generate() (fp32, host cores): `batch` sequences x `n_events` new events, EOS masked."""
    orc = _load_oracle()
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


# ------------------------------------------------------------------------------------------------------------------
# process set-up: spawning, rendezvous, loud failures
# ------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec this script under torch.distributed.run with N local ranks
    (one per GPU, rendezvous on 127.0.0.1) and pass its exit code on.  Refuses when the box has fewer than N GPUs."""
    if not (args.stub or args.emu):
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this box exposes {have} GPU(s); refusing to report an "
                             f"{args.gpus}-GPU number from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] spawning {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def setup(args):
    """-> (world, rank, local, dist or None).  The rank count must be what --gpus says: a mismatch is an error, never a
    silently relabelled run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values "
                         f"(or run `python bench.py --gpus {args.gpus}` alone and let it spawn its ranks)")
    import torch.distributed as dist
    if args.stub or args.emu:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return world, rank, local, (dist if world > 1 else None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU implementation)")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: local rank {local} has no GPU (device_count {torch.cuda.device_count()})")
    torch.cuda.set_device(local)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # (--force-dist rehearsal; with more ranks every rank must be given the SAME port by its launcher)
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        elif "MASTER_PORT" not in os.environ:
            raise SystemExit("bench.py: WORLD_SIZE > 1 needs MASTER_PORT in the environment (torch.distributed.run sets it)")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world
    return world, rank, local, (dist if (world > 1 or args.force_dist) else None)


def timed(fn_step, steps: int, dist, sync):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks"""
    gc.collect()  # (a full collection that is already due lands here rather than between two launches of the timed region)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        out = fn_step(i)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    timed.per_rank = [dt]
    if dist is not None:
        # every rank's own clock over the same barrier-bracketed region: the MAX is the job's time, the spread names a straggler
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu")
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)
        timed.per_rank = [float(x.item()) for x in every]
        dt = max(timed.per_rank)
    return dt, out


timed.per_rank = []


def multi_rank_fields(args, world: int, value: float, per_rank):
    """what an N > 1 line carries beyond the contract: every rank's own time over the timed region (the job's time is their max)
    and, given the 1-GPU number, the scaling efficiency (the driver computes its own from the per-N lines)"""
    out = {}
    multi = world > 1 or getattr(args, "force_dist", False)
    if multi and per_rank:
        out["ranks"] = per_rank_report(per_rank, args.steps)
    if multi and args.baseline_1gpu:
        out["baseline_1gpu"] = args.baseline_1gpu
        out["scaling_efficiency"] = value / (world * args.baseline_1gpu)
    return out


def per_rank_report(per_rank, steps: int):
    ms = [1e3 * x / steps for x in per_rank]
    return {"ms_per_step_by_rank": ms, "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
            "straggler_ratio": max(ms) / min(ms) if min(ms) > 0 else None}


def comm_info(world, dist):
    if dist is None:
        return {"world_size": 1, "backend": None}
    info = {"world_size": dist.get_world_size(), "backend": dist.get_backend()}
    if info["backend"] == "nccl":  # "nccl" IS RCCL on ROCm
        try:
            info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            pass
    return info


# ------------------------------------------------------------------------------------------------------------------
# kernel-family accounting from the per-launch HIP events (lib().profile)
# ------------------------------------------------------------------------------------------------------------------
FAMILIES = (
    ("gemm", ("mh_gemm",)),
    ("attention", ("mh_attn_",)),
    ("tokattn", ("mh_tokattn",)),
    ("optimizer", ("mh_adamw", "mh_sumsq", "mh_clip_coef")),
    ("loss", ("mh_cross_entropy", "mh_sum_f32", "mh_count_valid")),
    ("embedding", ("mh_embed", "mh_concat_tok", "mh_cast_from_f32", "mh_copy_rows")),
)


def family_of(name: str) -> str:
    for fam, prefixes in FAMILIES:
        if name.startswith(prefixes):
            return fam
    return "elementwise"  # rmsnorm, rope, swiglu, colsum, transpose


def summarize_launches(prof, wall_s: float, steps: int):
    fam, by_name = {}, {}
    for name, e0, e1 in prof:
        ms = e0.elapsed_time(e1)
        fam[family_of(name)] = fam.get(family_of(name), 0.0) + ms
        t, n = by_name.get(name, (0.0, 0))
        by_name[name] = (t + ms, n + 1)
    total = sum(fam.values())
    out = {k: {"ms_per_step": v / steps, "share_of_step": v * 1e-3 / wall_s} for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}
    out["other (torch glue, gaps, collectives)"] = {"ms_per_step": (wall_s * 1e3 - total) / steps,
                                                    "share_of_step": 1.0 - total * 1e-3 / wall_s}
    return out, by_name


def summarize_allreduce(windows, steps: int, exposed_ms=()):
    """windows: (bytes, launches) of every gradient exchange inside the timed region -- ONE per optimiser step, i.e. per
    accumulation window of `accumulate_grad_batches` timed steps (micro-batches; DDP no_sync inside a window).  Per optimiser
    step the payload is the whole flat gradient buffer (467,685,376 B for tv2o-medium in bf16); per timed step it is that
    divided by the window length."""
    n = len(windows)
    total = sum(b for b, _ in windows)
    return {"allreduce_windows": n,
            "allreduce_bytes_per_optimizer_step": (total / n) if n else 0,
            "allreduce_bytes_per_step": total / max(1, steps),
            "allreduce_launches_per_optimizer_step": (sum(l for _, l in windows) / n) if n else 0,
            "allreduce_exposed_ms_per_optimizer_step": (sum(exposed_ms) / len(exposed_ms)) if exposed_ms else None,
            "allreduce_exposed_ms_per_step": (sum(exposed_ms) / max(1, steps)) if exposed_ms else None}


def pmc_traffic(kind: str = "gemm"):
    """memory-side bytes per gemm_pp256 launch (kind "gemm", tools/gpu_pmc_bench.sh) or per generated event (kind "generate",
    tools/gpu_decode_profile.sh) from the newest committed rocprofv3 PMC pass; counters need their own rocprofv3 runs, so the
    bench line cites the file it read"""
    def run_key(path):  # r02_run23_... -> (2, 23), r05_final_... -> (5, 10**6), r06_final4_... -> (6, 10**6 + 4): newest by round and run number
        m = re.match(r"r(\d+)_(?:run(\d+)|final(\d*))_", os.path.basename(path))
        if not m:
            return (-1, -1)
        return (int(m.group(1)), int(m.group(2)) if m.group(2) else 10 ** 6 + int(m.group(3) or 0))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{kind}_traffic.json")), key=run_key)
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return d, os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


# ------------------------------------------------------------------------------------------------------------------
# mode: block (the north_star target)
# ------------------------------------------------------------------------------------------------------------------
def measure_block(args, B: int, S: int, steps: int, warmup: int, dist=None):
    """ONE event-level block forward (engine.layer_forward_folded: 7 launches; engine.layer_forward: 8) on a resident [B*S, 1024] bf16 input; FLOPs
    41,945,088 per event at S=4096 against the 2.5 PFLOP/s bf16 MFMA peak."""
    import midi_model_amd as mm
    from midi_model_amd import engine, ops
    from midi_model_amd.data import synthetic_events
    from midi_model_amd.lib import lib
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = mm.MIDIModel(mm.MIDIModelConfig.from_name(args.config)).to("cuda", dtype).eval()
    spec, W = model._specs["net"], model._W["net"]
    ev = synthetic_events(model.tokenizer, B, S, seed=7, device="cuda")
    rope = model.rope("net")
    rope.ensure(S)
    with torch.no_grad():
        x = torch.empty((B * S, spec.D), dtype=dtype, device="cuda")
        ops.embed_sum_fwd(ev.view(B * S, -1), W.embed, x)
        lw = W.layers[1]
        # The forward-only block as MIDIModel.forward runs it under no_grad from 8192 rows up (r06: the folded weight copies are kept
        # on the model, MIDIModel.folded_weights, re-derived only when a parameter changed): both RMSNorms folded around the
        # projections (engine.layer_forward_folded).  Block 1 is timed on what block 0 of the same path hands it: the residual
        # stream and its row statistics ([D / 64, M] partial sums of squares left by block 0's down projection).
        folded_form = (not args.block_save) and not args.block_unfolded and ops.norm_fold_ok(x, spec.D, spec.hd, spec.I)
        parts_in = None
        fold_ms = None
        if folded_form:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            folds = model.folded_weights("net")          # the one-off cost a first forward (or one after a weight update) pays
            torch.cuda.synchronize()
            fold_ms = 1e3 * (time.perf_counter() - t0)
            assert model.folded_weights("net") is folds  # (kept: nothing changed)
            fold = folds[1]
            x, parts_in = engine.layer_forward_folded(spec, W.layers[0], folds[0], x, B, S, rope, None)
        else:
            x, _ = engine.layer_forward(spec, W.layers[0], x, B, S, rope)   # input of block 1: a real residual stream

        def step(i):  # (save=False: the forward-only form a prompt prefill runs -- gate|up is not written for a backward)
            if folded_form:
                return engine.layer_forward_folded(spec, lw, fold, x, B, S, rope, parts_in)[0]
            return engine.layer_forward(spec, lw, x, B, S, rope, save=args.block_save)[0]

        # warm-up: W steps AND at least 0.5 s of this very work -- the part ramps its clocks over tens of milliseconds of load, and three
        # 2.8 ms blocks from idle were timed on the ramp (r05: per-launch minima 10 % under the averages inside one 13-step run)
        t_w = time.perf_counter()
        i = 0
        while i < warmup or time.perf_counter() - t_w < 0.5:
            step(i)
            i += 1
            if i % 8 == 0:
                torch.cuda.synchronize()
        warm_steps = i
        dt, _ = timed(step, steps, dist, torch.cuda.synchronize)
        # the same block with its two RMSNorm passes (the form below 8192 rows, and the r04 number), and the PUBLIC call:
        # MIDIModel.forward(events) under no_grad = embedding + 12 blocks + final norm, with the path it takes counted
        unfolded_ms = api = None
        if folded_form:
            for i in range(warmup):
                engine.layer_forward(spec, lw, x, B, S, rope, save=False)
            dtu, _ = timed(lambda i: engine.layer_forward(spec, lw, x, B, S, rope, save=False)[0], steps, dist, torch.cuda.synchronize)
            unfolded_ms = 1e3 * dtu / steps
            calls = {"folded": 0, "unfolded": 0}
            real_f, real_u = engine.layer_forward_folded, engine.layer_forward

            def count_f(*a, **k):
                calls["folded"] += 1
                return real_f(*a, **k)

            def count_u(*a, **k):
                calls["unfolded"] += 1
                return real_u(*a, **k)

            engine.layer_forward_folded, engine.layer_forward = count_f, count_u
            try:
                model(ev)
                n_f, n_u = calls["folded"], calls["unfolded"]
            finally:
                engine.layer_forward_folded, engine.layer_forward = real_f, real_u
            n_api = max(3, steps // 3)
            dta, _ = timed(lambda i: model(ev), n_api, dist, torch.cuda.synchronize)
            prof_api = []
            lib().profile = prof_api
            model(ev)
            torch.cuda.synchronize()
            lib().profile = None
            _, api_by_name = summarize_launches(prof_api, 1.0, 1)
            api_kern = {k: {"us_per_call": 1e3 * t / n, "calls": n} for k, (t, n) in sorted(api_by_name.items(), key=lambda kv: -kv[1][0])}
            api = {"call": "MIDIModel.forward(x) under torch.no_grad(), x = (16, 4096, 8) int64 event ids" if (B, S) == (16, 4096) else
                           f"MIDIModel.forward(x) under torch.no_grad(), x = ({B}, {S}, 8)",
                   "ms_per_call": 1e3 * dta / n_api, "blocks": spec.L, "folded_block_calls": n_f, "unfolded_block_calls": n_u,
                   "ms_per_block_incl_embedding_and_final_norm": 1e3 * dta / n_api / spec.L,
                   "frac_of_peak_whole_call": block_flops_per_event(S, spec.D, spec.I) * B * S * spec.L / (dta / n_api) / 1e12 / PEAK_BF16_TFLOPS,
                   "kernels": api_kern}
        # the per-kernel breakdown comes from a second pass with HIP events around every launch (the event markers
        # between the kernels cost the timed pass 1-2 %, as in train mode)
        prof = []
        lib().profile = prof
        dt_prof, _ = timed(step, steps, dist, torch.cuda.synchronize)
        lib().profile = None
    fl = block_flops_per_event(S, spec.D, spec.I) * B * S
    _, by_name = summarize_launches(prof, dt_prof, steps)
    ach = fl * steps / dt / 1e12
    kern = {k: {"us_per_call": 1e3 * t / n, "calls_per_block": n // steps} for k, (t, n) in sorted(by_name.items(), key=lambda kv: -kv[1][0])}
    attn_ms = sum(t for k, (t, n) in by_name.items() if k.startswith("mh_attn")) / steps
    gemm_ms = sum(t for k, (t, n) in by_name.items() if k.startswith("mh_gemm")) / steps
    attn_fl = 2.0 * spec.D * (S + 1) * B * S
    del model
    torch.cuda.empty_cache()
    return {
        "what": "one net block forward (LlamaDecoderLayer.forward, TF modeling_llama.py:295-324): RMSNorm, q|k|v + RoPE (projection "
                "epilogue), causal flash attention, o + residual, RMSNorm, gate|up + SwiGLU (projection epilogue), down + residual",
        "form": ("training forward (activations kept for the backward)" if args.block_save else
                 "forward only (prefill / validation: gate|up not stored)" + ("; both RMSNorms folded around the projections" if folded_form else "")),
        "norms_folded": bool(folded_form), "warmup_steps": warm_steps,
        "api_path": ("MIDIModel.forward under no_grad takes THIS form from 8192 rows up (folded weights kept on the model)" if folded_form
                     else "not the form MIDIModel.forward takes at this size (it folds the norms)"),
        "fold_ms_one_off": fold_ms, "ms_per_block_unfolded": unfolded_ms, "api_forward": api,
        "batch": B, "seq_len": S, "dtype": args.dtype, "ms_per_block": 1e3 * dt / steps, "events_per_s": B * S * steps / dt,
        "flops_per_event": block_flops_per_event(S, spec.D, spec.I),
        "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                     "target_frac": 0.40},
        "gemm_tflops": (fl - attn_fl) / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None,
        "attention_tflops": attn_fl / (attn_ms * 1e-3) / 1e12 if attn_ms else None,
        "launch_time_share": {"gemm": gemm_ms / (1e3 * dt_prof / steps), "attention": attn_ms / (1e3 * dt_prof / steps)},
        "ms_per_block_with_launch_events": 1e3 * dt_prof / steps,
        "kernels": kern,
    }


# ------------------------------------------------------------------------------------------------------------------
# mode: generate (BASELINE.json configs[3])
# ------------------------------------------------------------------------------------------------------------------
def measure_generate(args, world: int, rank: int, dist, steps: int, warmup: int):
    """generate(), batch 64, BOS prompt, 1024 new events, temp 1 / top_p 0.98 / top_k 20, seeded, EOS masked so that every
    row runs the full length (ban_eos; stated in config).  One step = one generate() call.  Replicas only across GPUs
    (SURVEY.md 8(e)): every rank generates its own batch, no collective."""
    import midi_model_amd as mm
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = mm.MIDIModel(mm.MIDIModelConfig.from_name(args.config)).to("cuda", dtype).eval()
    B, n_new = args.gen_batch, args.gen_events
    gen = torch.Generator(device="cuda")
    tok_steps = [0]

    def run(seed):
        gen.manual_seed(seed)
        with torch.no_grad():
            return model.generate(None, batch_size=B, max_len=1 + n_new, temp=1.0, top_p=0.98, top_k=20, generator=gen, ban_eos=True)

    def step(i):
        out = run(1000 * rank + i)
        assert out.shape == (B, 1 + n_new, 8), out.shape
        tok_steps[0] += int((out[:, 1:, :] != 0).any(axis=0).sum())  # token positions some row filled (lower bound of steps run)
        return out

    for i in range(warmup):  # full length: the decode session (K/V capacity, captured graphs) is sized by max_len
        run(100 + i)
    dt, _ = timed(step, steps, dist, torch.cuda.synchronize)
    per_event_s = dt / (n_new * steps)
    steps_per_event = tok_steps[0] / (n_new * steps)
    by = decode_bytes_per_event(B, (1 + n_new) / 2.0, steps_per_event)
    ses = model._sessions.idle[0] if model._sessions.idle else None
    nodes = getattr(ses, "nodes_per_event", None)
    del model
    torch.cuda.empty_cache()
    pmc, pmc_src = pmc_traffic("generate")
    pmc_ok = pmc is not None and (B, n_new) == (64, 1024)  # (the committed pass is of THIS workload; other shapes get none)
    return {
        "metric": f"MIDI events/sec, KV-cached generate(), {args.config}", "value": world * B * n_new * steps / dt, "unit": "events/s",
        "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
        "config": {"workload": f"{args.config} {args.dtype} generate(): batch {B} per GPU, BOS prompt, {n_new} new events, "
                               f"temp 1.0 top_p 0.98 top_k 20, seeded; EOS masked so every row runs the full length "
                               f"(BASELINE.json configs[3]); random-init weights",
                   "global_batch": world * B, "new_events": n_new, "parallelism": f"replicas x{world}",
                   "ms_per_event_step": 1e3 * per_event_s, "token_steps_per_event": steps_per_event,
                   "graph_nodes_per_event": nodes},
        "roofline": {"bound": "hbm", "kernel": "decode step (weights + K/V cache streamed once per event / token step)",
                     "achieved": by / per_event_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": by / per_event_s / (PEAK_HBM_GBS * 1e9),
                     "traffic": pmc["bytes_per_event_step"] if pmc_ok else None, "traffic_source": pmc_src if pmc_ok else None,
                     "traffic_detail": ({k: v for k, v in pmc.items() if k not in ("bytes_per_event_step", "top_readers_bytes_per_event")}
                                        if pmc_ok else None),
                     "algorithmic_bytes_per_event_step": by},
    }


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: "tv2o-large (2x hidden, 2x layers) bf16 ... seq_len=4096 events", 16 sequences per GPU.  Both readings
# (SURVEY.md 8(d) config 5): the reference's preset tv2o-large (24 + 6 layers, D = 1024) and the BASELINE-worded shape
# (24 + 6 layers, D = 2048, 32 heads, MLP 8192).  One GPU's share of the 8-GPU job = the same per-GPU step.
# ------------------------------------------------------------------------------------------------------------------
def measure_large(args, which: str, B: int, S: int, steps: int, warmup: int):
    import midi_model_amd as mm
    from midi_model_amd.data import synthetic_events
    from midi_model_amd.train import TrainMIDIModel
    if which == "tv2o-large":
        cfg = mm.MIDIModelConfig.from_name("tv2o-large")
        label = "tv2o-large (reference preset: 24 + 6 layers, D = 1024)"
    else:
        cfg = mm.MIDIModelConfig.get_config("v2", True, 24, 32, 2048, 8192)
        label = "2x-hidden large (BASELINE.json wording: 24 + 6 layers, D = 2048, 32 heads, MLP 8192)"
    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    t_build = time.perf_counter()
    try:  # parameters created and initialised on the device (1.8 G parameters take a minute of host time otherwise)
        with torch.device("cuda"):
            model = TrainMIDIModel(cfg, lr=2e-4, weight_decay=0.01, warmup=1e3, max_step=1e6, accumulate_grad_batches=1)
        model = model.to(torch.device("cuda"), torch.bfloat16)
    except Exception:
        model = TrainMIDIModel(cfg, lr=2e-4, weight_decay=0.01, warmup=1e3, max_step=1e6, accumulate_grad_batches=1)
        model = model.to(torch.device("cuda"), torch.bfloat16)
    model.configure_optimizers()
    # the 2x-hidden shape at 16 x 4096 per GPU peaked at 305.9e9 of the card's 309.2e9 bytes with every activation kept (r04): no
    # room for RCCL's channel buffers on a multi-GPU run.  Lean activation saving (SwiGLU outputs recomputed in the backward,
    # identical bits) takes ~39 GB off that peak.
    model.lean_activations = (which != "tv2o-large") or bool(os.environ.get("MH_LEAN_ACTIVATIONS"))
    t_build = time.perf_counter() - t_build
    n_params = sum(p.numel() for p in model.parameters())
    nc, tc = cfg.net_config, cfg.net_token_config
    batches = [synthetic_events(model.tokenizer, B, S + 1, seed=2000 + i, device="cuda") for i in range(2)]

    def step(i):
        return model.fit_step(batches[i % 2])

    tried = B
    while True:  # the largest per-GPU batch that fits, starting from the configured one
        try:
            for i in range(warmup):
                step(i)
            break
        except torch.OutOfMemoryError:
            del batches
            torch.cuda.empty_cache()
            if B == 1:
                raise
            B //= 2
            batches = [synthetic_events(model.tokenizer, B, S + 1, seed=2000 + i, device="cuda") for i in range(2)]
            model.zero_grad()
    dt, loss = timed(step, steps, None, torch.cuda.synchronize)
    fl = train_flops_per_event(S, net_L=nc.num_hidden_layers, tok_L=tc.num_hidden_layers, D=nc.hidden_size, I=nc.intermediate_size,
                               It=tc.intermediate_size, V=model.tokenizer.vocab_size)
    out = {"workload": f"{label}, bf16 training step (fwd+bwd+clip+AdamW), per-GPU batch {B} x {S} events (BASELINE.json configs[4] "
                       f"asks for 16 per GPU{'' if B == tried else f': {tried} does not fit 288 GB, this is the largest power of two that does'})",
           "value": B * S * steps / dt, "unit": "events/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
           "batch": B, "seq_len": S, "params": n_params, "loss": float(loss.item()),
           "train_flops_per_event": fl, "model_tflops": fl * B * S * steps / dt / 1e12,
           "model_flops_frac_of_peak": fl * B * S * steps / dt / 1e12 / PEAK_BF16_TFLOPS,
           "hbm_peak_gb": torch.cuda.max_memory_allocated() / 1e9,
           "hbm_headroom_gb": (torch.cuda.get_device_properties(0).total_memory - torch.cuda.max_memory_allocated()) / 1e9,
           "lean_activations": bool(model.lean_activations), "build_s": t_build}
    del model, batches
    gc.collect()
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------------
# mode: stub (spawn-path self test on CPU under gloo; never a measurement)
# ------------------------------------------------------------------------------------------------------------------
def run_stub(args):
    world, rank, local, dist = setup(args)
    x = torch.ones(1024)

    def step(i):
        y = x * (i + 1)
        if dist is not None:
            dist.all_reduce(y)
        return y

    for i in range(args.warmup):
        step(i)
    dt, y = timed(step, args.steps, dist, lambda: None)
    if rank == 0:
        out = {"metric": "STUB: spawn/rendezvous self-test of bench.py, NOT a measurement", "value": 0.0, "unit": "none",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "comm": comm_info(world, dist), "allreduce_sum_check": float(y[0])}
        out.update(multi_rank_fields(args, world, 1.0, timed.per_rank))
        print(json.dumps(out))
    if dist is not None:
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
    ap.add_argument("--no-gemm-events", action="store_true", help="no HIP events in the timed region (A/B runs)")
    ap.add_argument("--no-extras", action="store_true", help="train mode: skip the `block` and `generate` objects")
    ap.add_argument("--no-large", action="store_true", help="train mode: skip the `large` / `large_2x_hidden` objects (BASELINE configs[4])")
    ap.add_argument("--accumulate", type=int, default=1, help="accumulate_grad_batches (reference default 2; headline: 1)")
    ap.add_argument("--mode", default="train", choices=["train", "generate", "block", "large"],
                    help="train: the headline metric (BASELINE.json configs[1]); generate: KV-cached generate(), configs[3]; "
                         "block: one net block forward at --block-seq (north_star target)")
    ap.add_argument("--gen-batch", type=int, default=64)
    ap.add_argument("--gen-events", type=int, default=1024, help="new events per sequence per generate() call")
    ap.add_argument("--block-batch", type=int, default=16)
    ap.add_argument("--block-seq", type=int, default=4096)
    ap.add_argument("--block-unfolded", action="store_true",
                    help="block mode: the forward-only block with its two RMSNorm passes (the r04 form) instead of the folded norms")
    ap.add_argument("--block-save", action="store_true",
                    help="block mode: the TRAINING forward (also stores gate|up for the backward) instead of the forward-only form")
    ap.add_argument("--comm", default=os.environ.get("MH_COMM", "torch"), choices=["torch", "mh", "both"],
                    help="gradient exchange: torch = torch.distributed 'nccl' (RCCL, the default); mh = the library's own RCCL "
                         "communicator (mh_comm_*: one pre-multiplied-sum collective per bucket); both (N > 1) = the headline on "
                         "torch, then the same K steps on mh in the same allocation (`comm_ab` in the line)")
    ap.add_argument("--baseline-1gpu", type=float, default=None,
                    help="N > 1: the 1-GPU events/s of the same configuration; the line then carries scaling_efficiency = "
                         "value / (N * this)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # CPU/gloo self-test of the spawn path
    # CPU/gloo self-test of the N > 1 TRAIN path itself (parameter broadcast, bucketed reducer, per-rank clocks, the rank-0 line):
    # a tiny model on the tests' CPU stand-ins of the kernels (tests/emu_ops.py).  Never a measurement -- the metric says so.
    ap.add_argument("--emu", action="store_true", help=argparse.SUPPRESS)
    # One-GPU rehearsal of the N > 1 code path on the REAL backend: the process group is initialised ("nccl" = RCCL, device_id) and
    # every collective of the multi-GPU line runs at world size 1 -- barrier, the all-gather of the ranks' clocks, the parameter
    # broadcast, the bucketed gradient all-reduce inside fit_step, --comm both.  Launch under torch.distributed.run with ONE rank
    # (tests/test_ddp_gpu.py does); the line says what it is.  Not a scaling number.
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn(args))
    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL prints a version banner from
    # ncclCommInitRank), so file descriptor 1 is pointed at stderr for the whole run and Python's sys.stdout keeps the real one
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = real_stdout
    if args.stub:
        return run_stub(args)
    if args.emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_ops
        emu_ctx = emu_ops.install()
        emu_ctx.__enter__()   # (held until the process exits)
        args.mode, args.dtype = "train", "fp32"
        args.no_gemm_events = args.no_extras = args.no_cpu_baseline = True

    world, rank, local, dist = setup(args)
    if args.mode == "generate":
        g = measure_generate(args, world, rank, dist, args.steps, args.warmup)
        if rank == 0:
            out = {"metric": g["metric"], "value": g["value"], "unit": g["unit"], "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": g["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "config": g["config"], "roofline": g["roofline"],
                   "comm": comm_info(world, dist)}
            if world == 1 and not args.no_cpu_baseline:
                try:
                    out["cpu_baseline"] = cpu_baseline_generate(args.gen_batch, 32)  # B=64 for 32 events
                except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                    out["cpu_baseline"] = {"value": None, "unit": "events/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.mode == "large":
        res = {k: measure_large(args, w, args.batch, args.seq if args.seq != 2048 else 4096, args.steps, args.warmup)
               for k, w in (("large", "tv2o-large"), ("large_2x_hidden", "2x-hidden"))}
        if rank == 0:
            lg = res["large"]
            print(json.dumps({"metric": "MIDI events/sec, training step, tv2o-large, seq=4096 (BASELINE.json configs[4], one GPU's share)",
                              "value": lg["value"], "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": lg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": args.dtype, "data": "synthetic", "config": {"workload": lg["workload"]}, **res}))
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.mode == "block":
        b = measure_block(args, args.block_batch, args.block_seq, max(args.steps, 30), max(args.warmup, 3), dist)
        if rank == 0:
            print(json.dumps({"metric": f"MIDI events/sec through ONE net block forward, {args.config}, seq={args.block_seq}",
                              "value": world * b["events_per_s"], "unit": "events/s", "n_gpus": world, "steps": max(args.steps, 30),
                              "warmup": b["warmup_steps"], "ms_per_step": b["ms_per_block"], "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                              "config": {"workload": f"{args.config} {args.dtype} fused transformer-block forward, batch {args.block_batch} "
                                                     f"x {args.block_seq} events (north_star target: >= 0.40 of bf16 MFMA peak)"},
                              "roofline": b["roofline"], "block": b}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import midi_model_amd as mm
    from midi_model_amd import ops
    from midi_model_amd.data import synthetic_events
    from midi_model_amd.lib import lib
    from midi_model_amd.train import TrainMIDIModel

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    dev = torch.device("cpu") if args.emu else torch.device("cuda", local)
    sync = (lambda: None) if args.emu else torch.cuda.synchronize
    cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512) if args.emu else mm.MIDIModelConfig.from_name(args.config)
    model = TrainMIDIModel(cfg, lr=2e-4, weight_decay=0.01, warmup=1e3, max_step=1e6, accumulate_grad_batches=args.accumulate,
                           **({"bucket_mb": 4} if args.emu else {}))
    model = model.to(dev, dtype)
    model.configure_optimizers()
    multi = world > 1 or args.force_dist
    if args.force_dist:
        model.force_reduce = True   # (one rank: the bucketed exchange runs anyway, through the initialised process group)
    if args.comm == "mh" and multi:
        from midi_model_amd.comm import MHComm
        model.use_comm(MHComm.from_process_group(local))   # ("both": the headline runs on torch, the A/B leg follows below)
    model.broadcast_parameters(0)
    B, S = args.batch, args.seq
    batches = [synthetic_events(model.tokenizer, B, S + 1, seed=1000 + 17 * rank + i, device=dev) for i in range(2)]

    def step(i):
        return model.fit_step(batches[i % 2])  # (accumulate 1: training_step + optimizer_step every call)

    for i in range(args.warmup):
        step(i)
    prof = None if args.no_gemm_events else []
    ops.gemm_profile = prof  # HIP events around the projection GEMMs only (the roofline kernel) inside the timed region
    if model._reducer is not None:
        model._reducer.profile = True
        model._reducer.stats.clear()
    dt, loss = timed(step, args.steps, dist, sync)
    per_rank = list(timed.per_rank)
    ops.gemm_profile = None
    loss_v = float(loss.item())
    red = model._reducer
    # the reducer's statistics belong to the timed region ONLY: snapshot them and stop collecting before the extra passes below
    # (the un-instrumented re-run and the kernel-family steps would otherwise add their windows to the per-step numbers)
    red_stats = list(red.stats) if red is not None else []
    if red is not None:
        red.profile = False
    # the same K steps once more WITHOUT the HIP events around the GEMM launches (what the instrumentation of the timed
    # region costs is visible in the line: value_no_gemm_events beside value)
    dt_plain = None
    if not args.no_gemm_events:
        dt_plain, _ = timed(step, args.steps, dist, torch.cuda.synchronize)
    # kernel-family shares: events around EVERY C-ABI launch cost ~2 % of the step (the host falls behind on the short
    # kernels), so they are taken over two extra steps AFTER the timed region and normalised by those steps' own wall time
    launches, fam_steps, fam_dt = None, 2, None
    if not args.no_gemm_events:
        launches = []
        lib().profile = launches
        fam_dt, _ = timed(step, fam_steps, dist, torch.cuda.synchronize)
        lib().profile = None

    # --comm both (N > 1): the same K steps once more with the buckets going out through the library's own communicator, in the
    # same allocation.  Every rank reports whether its communicator came up BEFORE any collective is issued on it: a rank
    # that failed alone would otherwise leave the others waiting in their first bucket.
    comm_ab = None
    if multi and args.comm == "both" and not args.emu:
        ok, err = 1, None
        try:
            from midi_model_amd.comm import MHComm
            mh = MHComm.from_process_group(local)
        except Exception as e:
            ok, err, mh = 0, repr(e), None
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            model.use_comm(mh)
            for i in range(2):
                step(i)
            dt_mh, _ = timed(step, args.steps, dist, torch.cuda.synchronize)
            comm_ab = {"torch_ms_per_step": 1e3 * (dt_plain if dt_plain else dt) / args.steps, "mh_ms_per_step": 1e3 * dt_mh / args.steps,
                       "mh_ranks": per_rank_report(timed.per_rank, args.steps), "rccl_version_code": mh.rccl_version,
                       "what": "same model, same batches, K steps each: torch.distributed 'nccl' (divide + all-reduce(SUM) per bucket) "
                               "against mh_comm_allreduce (one pre-multiplied-sum all-reduce per bucket)"}
            model.comm = None
            model._reducer = None
            mh.close()
        else:
            comm_ab = {"error": err or "another rank could not open the library communicator"}
            if mh is not None:
                mh.close()

    # One GPU cannot run a ring, but it can run the exchange's CALL SEQUENCE: the same step with every bucket of the flat
    # gradient buffer sent through the library's RCCL communicator at world size 1 (15 x 32 MB launches on the communication
    # stream while the backward owns the CUs; with one rank RCCL's kernel is a scaled copy of the bucket, not a ring).  The
    # step-time delta is what those launches cost the compute stream here.
    contention = None
    if world == 1 and not multi and not args.no_extras and dt_plain is not None:
        try:
            from midi_model_amd.comm import MHComm
            model.force_reduce = True
            model.use_comm(MHComm(0, 1, local))
            for i in range(2):
                step(i)
            red_x = model._reducer
            red_x.profile = True
            red_x.stats.clear()
            dt_x, _ = timed(step, args.steps, dist, torch.cuda.synchronize)
            stx = list(red_x.stats)
            exposed_x = [a.elapsed_time(b) for a, b, _, _ in stx if a is not None]
            contention = {"ms_per_step_with_exchange": 1e3 * dt_x / args.steps, "ms_per_step_without": 1e3 * dt_plain / args.steps,
                          "allreduce_contention_ms": 1e3 * (dt_x - dt_plain) / args.steps,
                          "bytes_per_step": (sum(x[2] for x in stx) / len(stx)) if stx else None,
                          "launches_per_step": (sum(x[3] for x in stx) / len(stx)) if stx else None,
                          "exposed_ms_per_step": (sum(exposed_x) / len(exposed_x)) if exposed_x else None,
                          "rccl_version_code": model.comm.rccl_version,
                          "what": "world size 1 through mh_comm_allreduce (RCCL one-rank kernel per 32 MB bucket on the communication "
                                  "stream, overlapping the backward): the call sequence and stream overlap of the N-GPU step, NOT a "
                                  "ring over xGMI"}
            model.comm.close()
        except Exception as e:  # a probe must never cost the headline number
            contention = {"error": repr(e)}
        model.force_reduce = False
        model.comm = None

    if rank == 0:
        events = world * B * S * args.steps
        value = events / dt
        nc, tc = cfg.net_config, cfg.net_token_config
        fl_event = train_flops_per_event(S, net_L=nc.num_hidden_layers, tok_L=tc.num_hidden_layers, D=nc.hidden_size,
                                         I=nc.intermediate_size, It=tc.intermediate_size, V=model.tokenizer.vocab_size)
        out = {
            "metric": (f"MIDI events/sec, training step (fwd+bwd+clip+AdamW), {args.config}, seq={S}" if not args.emu else
                       "EMU: CPU self-test of bench.py's N > 1 train path on tests/emu_ops.py, NOT a measurement"),
            "value": value, "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.config} {args.dtype} training step, per-GPU batch {B} x {S} events x 8 tokens "
                                   f"({'BASELINE.json configs[1]' if (args.config, B, S) == ('tv2o-medium', 16, 2048) else 'non-headline configuration'}); "
                                   f"random-init weights, synthetic events",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}",
                       "accumulate_grad_batches": args.accumulate,
                       "optimizer": "AdamW bf16-true + global-norm clip 1.0" if args.dtype == "bf16" else "AdamW fp32 + clip"},
            "comm": comm_info(world, dist),
            "loss": loss_v,
            "value_no_gemm_events": (events / dt_plain) if dt_plain else None,
            "ms_per_step_no_gemm_events": (1e3 * dt_plain / args.steps) if dt_plain else None,
            "model_tflops_per_gpu": fl_event * B * S * args.steps / dt / 1e12,
            "model_flops_frac_of_peak": fl_event * B * S * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS,
        }
        if contention is not None:
            out["allreduce_contention"] = contention
            out["allreduce_contention_ms"] = contention.get("allreduce_contention_ms")
        out.update(multi_rank_fields(args, world, value, per_rank))
        if comm_ab is not None:
            out["comm_ab"] = comm_ab
        if args.force_dist:
            out["metric"] = "REHEARSAL (--force-dist, one rank on the real backend; not a scaling number): " + out["metric"]
        if multi:
            out["comm"]["exchange"] = "torch" if args.comm == "both" else args.comm
            st = red_stats
            exposed = [a.elapsed_time(b) for a, b, _, _ in st if a is not None]
            out.update(summarize_allreduce([(x[2], x[3]) for x in st], args.steps, exposed))
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
                w_ = sorted(1e3 * e0.elapsed_time(e1) for e0, e1, _, s_ in prof if s_ == shp)
                print(f"[bench]   {shp}: {n_:4d} {t_:9.3f} {f_ / (t_ * 1e-3) / 1e12:8.1f}   (us per launch: min {w_[0]:.1f}, median {w_[len(w_) // 2]:.1f}, max {w_[-1]:.1f})",
                      file=sys.stderr)
            if os.environ.get("MH_BENCH_WINDOWS"):  # the windows of one shape in launch order
                for shp in by_shape:
                    if str(shp) == os.environ["MH_BENCH_WINDOWS"]:
                        print("[bench]   windows in order:", " ".join(f"{1e3 * e0.elapsed_time(e1):.0f}" for e0, e1, _, s_ in prof if s_ == shp), file=sys.stderr)
            pmc, pmc_src = pmc_traffic()
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_pp256_kernel (all projection GEMMs: fwd, dgrad, wgrad, lm_head; split-K reductions inside the window)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                               "traffic": (pmc or {}).get("bytes_per_launch"), "traffic_source": pmc_src,
                               "traffic_detail": {k: v for k, v in (pmc or {}).items() if k != "bytes_per_launch"} or None,
                               "launches": n, "avg_launch_us": 1e3 * ms / n,
                               "avg_flops_per_launch": fl / n, "gemm_share_of_step_time": ms * 1e-3 / dt,
                               # the same ratio over the launches whose epilogue carries no SwiGLU forward / backward
                               "achieved_plain_epilogue": (sum(f for _, f in plain) / (sum(t for t, _ in plain) * 1e-3) / 1e12
                                                           if plain else None),
                               "launches_plain_epilogue": len(plain)}
        if launches:
            fams, by_name = summarize_launches(launches, fam_dt, fam_steps)
            out["kernel_families"] = fams
            out["kernel_families_note"] = (f"from {fam_steps} extra steps after the timed region with HIP events around every launch "
                                           f"({1e3 * fam_dt / fam_steps:.2f} ms/step with that instrumentation)")
            H, hd, L = nc.num_attention_heads, nc.hidden_size // nc.num_attention_heads, nc.num_hidden_layers
            fwd_fl = 4.0 * hd * S * (S + 1) / 2 * B * H * L * fam_steps          # QK^T + PV on the lower triangle
            fwd_ms = sum(by_name.get(k, (0.0, 0))[0] for k in ("mh_attn_fwd", "mh_attn_prep_fwd"))
            bwd_ms = sum(by_name.get(k, (0.0, 0))[0] for k in ("mh_attn_bwd", "mh_attn_bwd_o", "mh_attn_bwd_o_scaled", "mh_attn_prep_bwd"))
            out["attention"] = {"what": "event-level causal flash attention, head_dim 64 (prep kernels included)",
                                "fwd_us_per_layer": 1e3 * fwd_ms / (L * fam_steps), "bwd_us_per_layer": 1e3 * bwd_ms / (L * fam_steps),
                                "fwd_tflops": fwd_fl / (fwd_ms * 1e-3) / 1e12 if fwd_ms else None,
                                "bwd_tflops": 2.5 * fwd_fl / (bwd_ms * 1e-3) / 1e12 if bwd_ms else None,
                                "fwd_frac_of_peak": fwd_fl / (fwd_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if fwd_ms else None,
                                "bwd_frac_of_peak": 2.5 * fwd_fl / (bwd_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if bwd_ms else None}
    del model, batches
    if not args.emu:
        torch.cuda.empty_cache()
    if world == 1 and not multi and not args.no_extras:
        # the other two measurements the judge asks for, in the same driver-run line (N=1 only: replicas add nothing)
        for key, fn in (("block", lambda: measure_block(args, args.block_batch, args.block_seq, 30, 3)),
                        ("generate", lambda: measure_generate(args, 1, 0, None, 5, 1)),
                        ("large", lambda: measure_large(args, "tv2o-large", 16, 4096, 5, 1)),
                        ("large_2x_hidden", lambda: measure_large(args, "2x-hidden", 16, 4096, 5, 1))):
            if key.startswith("large") and args.no_large:
                continue
            try:
                out[key] = fn()
            except Exception as e:  # an extra must never cost the headline number
                out[key] = {"error": repr(e)}
        if rank == 0 and not args.no_cpu_baseline and isinstance(out.get("generate"), dict) and "error" not in out["generate"]:
            try:  # the CPU oracle's generate() beside the GPU number (bounded sample: batch 64 x 8 events)
                out["generate"]["cpu_baseline"] = cpu_baseline_generate(args.gen_batch, 32)  # batch 64 for 32 events
            except Exception as e:
                out["generate"]["cpu_baseline"] = {"value": None, "unit": "events/s", "cores": os.cpu_count(), "kind": "port",
                                                   "sample": f"failed: {e!r}"}
    if rank == 0:
        if world == 1 and not multi and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_seq)
                out["cpu_baseline"]["config0_forward"] = cpu_baseline_config0()
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "events/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
