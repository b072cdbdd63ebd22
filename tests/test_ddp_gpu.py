"""The data-parallel step's GPU branch on real HIP tensors and kernels: two ranks sharing the one GPU of the test box,
``gloo`` as the transport (RCCL refuses two ranks on one device; the reducer is backend-agnostic: the same bucket launches on
the communication stream, the same async work handles, the same stream waits).  What the CPU test (test_abi_and_ddp.py)
checks through the fake backend is checked here on the device: gradients stay local inside an accumulation window, the window's
close ships the flat buffer in buckets while the backward is still running, both ranks end with the SAME averaged gradient
(= the mean of what each computes alone), the same updated weights, and averaged validation metrics."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

import midi_model_amd as mm

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from conftest import load_oracle
    from midi_model_amd.train import TrainMIDIModel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = load_oracle()
    tok = mm.MIDITokenizerV2()
    cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512)
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    dev = torch.device("cuda", 0)
    batches = [orc.synthetic_events(tok, 2, 33, seed=100 + 10 * mb + rank).to(dev) for mb in range(2)]
    # what this rank computes alone over its window (no exchange): a second model whose reducer is switched off
    alone = TrainMIDIModel(cfg, lr=1e-2, warmup=0, accumulate_grad_batches=2).to(dev, torch.bfloat16)
    alone.load_state_dict(sd)
    alone._reducer_for_step = lambda: None
    for b in batches:
        alone.training_step(b)
    g_alone = alone.grad_buffer().float().cpu().numpy()

    model = TrainMIDIModel(cfg, lr=1e-2, warmup=0, accumulate_grad_batches=2, bucket_mb=1).to(dev, torch.bfloat16)
    if rank == 0:
        model.load_state_dict(sd)
    model.configure_optimizers()
    model.broadcast_parameters(0)  # rank 1 started from different random weights
    l0 = model.training_step(batches[0])          # inside the window: no exchange
    g_local = model.grad_buffer().float().cpu().numpy()
    l1 = model.training_step(batches[1])          # closes the window: bucketed all-reduce on the communication stream
    red = model._reducer
    assert red is not None and red.comm_stream is not None and len(red.works) > 0, "the reducer's GPU branch did not run"
    n_buckets = len(red.launched)
    red.profile = True
    red.finish()
    (ev0, ev1, nbytes, nlaunch), = red.stats
    assert nbytes == model._flat.numel() * model._flat.element_size() and n_buckets <= nlaunch <= n_buckets + 1
    torch.cuda.synchronize()
    exposed_ms = ev0.elapsed_time(ev1)            # HIP events around the wait for the communication stream
    g_avg = model.grad_buffer().float().cpu().numpy()
    model.optimizer_step()
    vloss, vacc = model.validation_step(batches[0])
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), g_alone=g_alone, g_local=g_local, g_avg=g_avg,
             flat=model._flat.detach().float().cpu().numpy(), n_buckets=n_buckets, vloss=vloss.float().cpu().numpy(),
             vacc=float(vacc), losses=np.array([l0.item(), l1.item()]), exposed_ms=exposed_ms)
    dist.destroy_process_group()


def test_ddp_world2_on_the_device(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert not np.allclose(r0["g_local"], r1["g_local"])            # inside the window gradients stayed local
    np.testing.assert_array_equal(r0["g_avg"], r1["g_avg"])          # after it both ranks hold the same bits
    assert int(r0["n_buckets"]) >= 3, "the flat buffer must have gone out in several 1 MB+ buckets"
    expect = (r0["g_alone"] + r1["g_alone"]) / 2                      # the mean of what each rank computes alone
    err = np.abs(r0["g_avg"] - expect)
    scale = np.sqrt(np.mean(expect ** 2))
    assert err.max() <= 2 ** -7 * np.abs(expect).max() + 1e-3 * scale, (err.max(), scale)   # bf16: pre-division + sum roundings
    assert np.sqrt(np.mean(err ** 2)) < 5e-3 * scale
    np.testing.assert_array_equal(r0["flat"], r1["flat"])            # identical weights after broadcast + identical update
    np.testing.assert_allclose(r0["vloss"], r1["vloss"], rtol=0, atol=0)
    assert r0["vacc"] == r1["vacc"]
    assert np.isfinite(r0["losses"]).all() and np.isfinite(r1["losses"]).all() and float(r0["exposed_ms"]) >= 0.0


def _worker_world1(rank, port, out_dir, backend):
    """the REAL RCCL backends at world size 1, forced through the reducer's whole bucketed path inside the benchmarked step
    (tv2o-medium bf16, 16 x 2048 events): `torch` = torch.distributed "nccl" (the default exchange), `mh` = the library's own
    communicator (mh_comm_*, one pre-multiplied-sum collective per bucket)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from midi_model_amd.comm import MHComm
    from midi_model_amd.data import synthetic_events
    from midi_model_amd.train import TrainMIDIModel
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    cfg = mm.MIDIModelConfig.from_name("tv2o-medium")
    torch.manual_seed(0)
    model = TrainMIDIModel(cfg, lr=2e-4, warmup=0, accumulate_grad_batches=1).to(dev, torch.bfloat16)
    model.configure_optimizers()
    start = model._flat.detach().clone()
    batch = synthetic_events(model.tokenizer, 16, 2049, seed=5, device="cuda")
    # the plain single-GPU step (no exchange)
    loss_plain = model.fit_step(batch).item()
    flat_plain = model._flat.detach().clone()
    # the same step from the same weights with the exchange forced through the real backend
    model._flat.copy_(start)
    model._opt["m"].zero_()
    model._opt["v"].zero_()
    model.global_step = 0
    model.force_reduce = True
    if backend == "mh":
        model.use_comm(MHComm.from_process_group(0))
        assert model.comm.world == 1 and model.comm.rccl_version > 20000
    model.broadcast_parameters(0)
    red_probe = model._reducer_for_step()
    assert red_probe is not None and red_probe.force and (red_probe.comm is not None) == (backend == "mh")
    red_probe.profile = True
    loss_x = model.fit_step(batch).item()
    torch.cuda.synchronize()
    (ev0, ev1, nbytes, nlaunch), = red_probe.stats
    # bit-identical -- except that the embedding tables' gradient rows are sums over an id's occurrences in an order the counting
    # sort's atomics leave unspecified (mh_token_segments; torch's own embedding backward is order-free in the same way): one or
    # two of 7 M embedding weights may land one bf16 step apart between two runs of the SAME step (tools/gpu_r06_determinism.py:
    # 0-1 differing gradient elements between repeats, always inside net.embed_tokens.weight, under every kernel option)
    diff = model._flat != flat_plain
    inside = torch.zeros_like(diff)
    for n in ("net.embed_tokens.weight", "net_token.embed_tokens.weight"):
        off, cnt, _ = model._offsets[n]
        inside[off:off + cnt] = True
    n_diff, n_outside = int(diff.sum()), int((diff & ~inside).sum())
    rel = ((model._flat.float() - flat_plain.float()).abs() / flat_plain.float().abs().clamp_min(1e-30))[diff]
    same = bool(n_outside == 0 and n_diff <= 8 and (n_diff == 0 or float(rel.max()) <= 2.0 ** -6))
    np.savez(os.path.join(out_dir, f"world1_{backend}.npz"), loss_plain=loss_plain, loss_x=loss_x, same=same, nbytes=nbytes, n_diff=n_diff,
             nlaunch=nlaunch, expect_bytes=model._flat.numel() * model._flat.element_size(), exposed_ms=ev0.elapsed_time(ev1),
             maxdiff=float((model._flat.float() - flat_plain.float()).abs().max()))
    if model.comm is not None:
        model.comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["torch", "mh"])
def test_rccl_world1_bucketed_exchange_inside_the_benchmarked_step(tmp_path, backend):
    """What one GPU can prove about the multi-GPU path: the real RCCL backends initialise, every bucket of the flat gradient
    buffer (467.7 MB of bf16 gradients, 32 MB buckets, back to front, on the communication stream while the backward runs)
    goes through them inside a real 16 x 2048 step, the coverage check passes, and with one rank the averaged gradient is the
    gradient: the step ends on the same bits as the plain single-GPU step (up to the embedding gradient's unspecified summation
    order: see the worker)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_world1, args=(_free_port(), str(tmp_path), backend), nprocs=1, join=True)
    r = np.load(tmp_path / f"world1_{backend}.npz")
    assert int(r["nbytes"]) == int(r["expect_bytes"]) == 467_685_376
    assert 14 <= int(r["nlaunch"]) <= 20, int(r["nlaunch"])
    assert float(r["loss_plain"]) == float(r["loss_x"])
    assert bool(r["same"]), (float(r["maxdiff"]), int(r["n_diff"]))
    assert float(r["exposed_ms"]) >= 0.0


def test_bench_multi_gpu_code_path_rehearsal_on_the_real_backend():
    """The N > 1 lines of bench.py on the backend the 8-GPU run will use, with ONE rank (`--force-dist`, launched exactly as the
    driver launches N ranks: `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 ...`): env:// rendezvous on
    127.0.0.1, init_process_group("nccl", device_id), the barrier + synchronize bracket, the all-gather of the ranks' clocks on
    the device, the parameter broadcast, fit_step with the bucketed all-reduce through torch.distributed's RCCL (15 x 32 MB on the
    communication stream), `--comm both` (the library's communicator built from the process group, then the same steps through
    mh_comm_allreduce), the all-reduce summary and the rank-0 JSON line.  What one rank cannot show is a ring; everything that
    can fail before the first ring starts runs here (train.py:461-474)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--comm", "both",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--baseline-1gpu", "300000"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("REHEARSAL") and d["n_gpus"] == 1 and d["steps"] == 2
    assert d["comm"]["backend"] == "nccl" and d["comm"]["world_size"] == 1 and d["comm"]["exchange"] == "torch"
    assert len(d["ranks"]["ms_per_step_by_rank"]) == 1 and abs(d["ranks"]["ms_per_step_max"] - d["ms_per_step"]) < 1e-6
    assert abs(d["scaling_efficiency"] - d["value"] / 300000.0) < 1e-9
    # one exchange of the whole bf16 gradient buffer per step, ~15 buckets, through torch's RCCL
    assert d["allreduce_windows"] == 2 and d["allreduce_bytes_per_optimizer_step"] == 467_685_376
    assert 14 <= d["allreduce_launches_per_optimizer_step"] <= 20
    ab = d["comm_ab"]
    assert "error" not in ab, ab
    assert ab["mh_ms_per_step"] > 0 and ab["torch_ms_per_step"] > 0 and abs(ab["mh_ms_per_step"] / ab["torch_ms_per_step"] - 1) < 0.1
    assert "roofline" in d and 0.2 < d["roofline"]["frac"] < 0.7 and "block" not in d and "cpu_baseline" not in d


@pytest.mark.parametrize("mode,extra", [("block", ["--block-batch", "2", "--block-seq", "512"]),
                                        ("block", ["--block-batch", "2", "--block-seq", "512", "--block-unfolded"]),
                                        ("generate", ["--gen-batch", "4", "--gen-events", "12"])],
                         ids=["block_folded", "block_unfolded", "generate"])
def test_bench_modes_print_one_valid_line(mode, extra):
    """the other modes of bench.py (the `block` and `generate` objects of the driver's line come from the same functions) at toy
    sizes: one JSON line on stdout with the contract's keys and a roofline object; the folded and unfolded block forms both run"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", mode, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["roofline"]["frac"] > 0
    if mode == "block":
        assert d["block"]["norms_folded"] == ("--block-unfolded" not in extra)
        names = set(d["block"]["kernels"])
        assert ("mh_gemm_rowss" in names) == d["block"]["norms_folded"] and ("mh_rmsnorm_fwd" in names) != d["block"]["norms_folded"]
