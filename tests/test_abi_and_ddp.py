"""(1) The C-ABI library loads on a GPU-less box and exports every symbol include/midihip.h declares (no compute
calls).  (2) The data-parallel path (world_size 2, gloo, CPU): bucketed overlapped gradient averaging, the
no_sync behaviour inside an accumulation window, parameter broadcast, validation metric sync."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

import midi_model_amd as mm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from midi_model_amd import lib as L
    import midi_model_amd.build as build
    build.build()
    protos = L.parse_header()
    assert len(protos) >= 34 and "mh_gemm_nt" in protos and "mh_attn_bwd" in protos, sorted(protos)
    handle = L.lib()  # raises if a declared symbol is missing from the .so
    for name in protos:
        assert hasattr(handle.cdll, name)
    assert handle.cdll.mh_version() >= 1
    # argument validation happens before any device work: a bad call fails loudly with a message
    with pytest.raises(RuntimeError, match="gemm"):
        handle.call("mh_gemm_nt", 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0, 0.0, 1, 1, 0, 0)


def test_ab_library_exports_the_same_abi_and_is_marked():
    """libmidihip_ab.so (build.py, -DMH_AB_BUILDS: the forms kept for comparisons) has the production ABI and says what it is;
    the package's own handle is the production library"""
    from midi_model_amd import lib as L
    import midi_model_amd.build as build
    build.build()
    assert L.lib().cdll.mh_ab_builds() == 0 and L.lib().path.endswith("libmidihip.so")
    with L.use_ab() as ab:
        assert L.lib() is ab and ab.cdll.mh_ab_builds() == 1
        for name in L.parse_header():
            assert hasattr(ab.cdll, name)
    assert L.lib().cdll.mh_ab_builds() == 0


def test_options_are_per_thread():
    """mh_set_option is a thread-local context: what one host thread selects (a test, a probe) is invisible to the launches of
    every other thread of the process (concurrent generators, app.py:496), and a new thread starts from the defaults"""
    import threading
    from midi_model_amd import lib as L
    h = L.lib()
    defaults = {n: h.cdll.mh_get_option(n.encode()) for n in ("gemm", "gemm_k64", "gemm_ablate", "skinny_mb", "skinny_nbt", "attn_v3", "attn_v3_wps")}
    assert defaults["gemm"] == 1 and defaults["attn_v3"] == 255 and defaults["gemm_ablate"] == 0
    seen = {}

    def other():
        seen["before"] = {n: h.cdll.mh_get_option(n.encode()) for n in defaults}
        h.call("mh_set_option", b"attn_v3", 15)
        seen["own"] = h.cdll.mh_get_option(b"attn_v3")

    try:
        h.call("mh_set_option", b"skinny_mb", 2)
        h.call("mh_set_option", b"attn_v3", 31)
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen["before"] == defaults, (seen["before"], defaults)
        assert seen["own"] == 15
        assert h.cdll.mh_get_option(b"attn_v3") == 31 and h.cdll.mh_get_option(b"skinny_mb") == 2
    finally:
        h.call("mh_set_option", b"skinny_mb", 0)
        h.call("mh_set_option", b"attn_v3", 255)
    assert h.cdll.mh_get_option(b"no_such_option") == -1


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    from midi_model_amd import lib as L
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(RuntimeError, match="only implementation"):
        L.lib()


def test_header_is_plain_c_and_a_c_host_can_call_the_library(tmp_path):
    """include/midihip.h is the boundary for hosts in ANY language: it must compile as strict C99 (no C++isms, no torch types),
    and a C program must be able to open the library, find every declared symbol and use the error / option entry points."""
    import shutil
    import subprocess
    from midi_model_amd import lib as L
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = str(tmp_path / "host")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_host", "host.c"), "-o", exe, "-ldl"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    names = sorted(L.parse_header())
    for path, ab in ((L.LIB_PATH, 0), (L.AB_LIB_PATH, 1)):
        r = subprocess.run([exe, path] + names, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
        assert f"A/B library {ab}" in r.stdout and f"{len(names)} symbols" in r.stdout, r.stdout


def test_build_then_use_keeps_one_hip_runtime_in_the_process():
    """build() in a fresh interpreter (nothing imported yet) must leave exactly one libamdhip64 / ROCr mapped: PyTorch-ROCm ships
    its own copies, and a library loaded BEFORE torch binds to /opt/rocm's -- the second runtime to initialise then reports
    "no ROCm-capable device is detected" at the first launch (seen on the GPU box with build() followed by smoke())."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import __graft_entry__ as g\n"
            "g.build()\n"
            "import torch\n"
            "m = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime64' in l))\n"
            "print('MAPPED', len([x for x in m if 'libamdhip64' in x]), len([x for x in m if 'libhsa-runtime64' in x]), m)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("MAPPED")][-1]
    assert line.split()[1:3] == ["1", "1"], line


def test_grad_reducer_bucketing_single_process():
    from midi_model_amd.train import GradReducer
    flat = torch.arange(100, dtype=torch.float32)
    r = GradReducer(flat, None, bucket_bytes=4 * 30)
    assert r.world == 1
    r.ready(80, 100)
    r.finish()
    assert torch.equal(flat, torch.arange(100, dtype=torch.float32))  # world 1: untouched


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from conftest import load_oracle
    import emu_ops
    from midi_model_amd.train import TrainMIDIModel
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = load_oracle()
    tok = mm.MIDITokenizerV2()
    cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512)
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    with emu_ops.install():
        model = TrainMIDIModel(cfg, lr=1e-2, warmup=0, accumulate_grad_batches=2, bucket_mb=1)
        if rank == 0:
            model.load_state_dict(sd)
        model.broadcast_parameters(0)  # rank 1 started from different random weights
        batches = [orc.synthetic_events(tok, 2, 9, seed=100 + 10 * mb + rank) for mb in range(2)]
        l0 = model.training_step(batches[0])          # inside the window: no exchange
        g_local = model.grad_buffer().clone()
        l1 = model.training_step(batches[1])          # closes the window: bucketed all-reduce
        n_buckets = len(model._reducer.launched)
        model._reducer.profile = True
        model._reducer.finish()
        (ev0, ev1, nbytes, nlaunch), = model._reducer.stats  # (no HIP events on CPU tensors; bytes / launches still counted)
        assert ev0 is None and nbytes == model._flat.numel() * model._flat.element_size() and n_buckets <= nlaunch <= n_buckets + 1
        g_avg = model.grad_buffer().clone()
        model.optimizer_step()
        vloss, vacc = model.validation_step(batches[0])
        # LoRA on the same ranks (train.py:439-449): only the adapter gradients are exchanged, in optimizer_step
        lora = TrainMIDIModel(cfg, lr=1e-2, warmup=0, accumulate_grad_batches=1)
        lora.load_state_dict(sd)
        lo = lora.add_adapter(r=8, lora_alpha=16, generator=torch.Generator().manual_seed(3))
        for name in lo.B:
            lo.B[name].fill_(0.01)
        lo.dirty = True
        lora.training_step(batches[0])
        assert lora._reducer is None
        lo.compute_grads(lora)
        lg_local = lo.grad.clone()
        lora.optimizer_step()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), g_local=g_local.numpy(), g_avg=g_avg.numpy(),
                 flat=model._flat.detach().numpy(), n_buckets=n_buckets, vloss=vloss.numpy(), vacc=float(vacc),
                 losses=np.array([l0.item(), l1.item()]), lora_g_local=lg_local.numpy(), lora_g=lo.grad.numpy(),
                 lora_flat=lo.flat.numpy())
    dist.destroy_process_group()


def test_ddp_world2_gloo(tmp_path, orc):
    import torch.multiprocessing as mp
    import emu_ops
    from midi_model_amd.train import TrainMIDIModel
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # inside the window gradients stayed local (they differ); after it both ranks hold the same average
    assert not np.allclose(r0["g_local"], r1["g_local"])
    np.testing.assert_array_equal(r0["g_avg"], r1["g_avg"])
    assert int(r0["n_buckets"]) >= 3, "the flat buffer (22 MB) must have gone out in several 1 MB+ buckets"
    # the average equals the mean of what each rank computes alone over its two micro-batches
    tok = mm.MIDITokenizerV2()
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    expect = 0
    with emu_ops.install():
        for rank in range(2):
            m = TrainMIDIModel(mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512), accumulate_grad_batches=2)
            m.load_state_dict(sd)
            for mb in range(2):
                m.training_step(orc.synthetic_events(tok, 2, 9, seed=100 + 10 * mb + rank))
            expect = expect + m.grad_buffer().numpy() / 2
    np.testing.assert_allclose(r0["g_avg"], expect, rtol=1e-4, atol=1e-7)
    # identical weights after broadcast + identical update; validation metrics were averaged across ranks
    np.testing.assert_array_equal(r0["flat"], r1["flat"])
    np.testing.assert_array_equal(r0["vloss"], r1["vloss"])
    assert float(r0["vacc"]) == float(r1["vacc"])
    # LoRA: local adapter gradients differ, the exchanged ones are their mean, the adapters stay in step
    assert not np.allclose(r0["lora_g_local"], r1["lora_g_local"])
    np.testing.assert_array_equal(r0["lora_g"], r1["lora_g"])
    np.testing.assert_allclose(r0["lora_g"], (r0["lora_g_local"] + r1["lora_g_local"]) / 2, rtol=1e-5, atol=1e-8)
    np.testing.assert_array_equal(r0["lora_flat"], r1["lora_flat"])


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher must run TWO ranks (it used to run one and label it 2): the spawn path
    re-execs under torch.distributed.run; --stub swaps the GPU step for a gloo all-reduce so the path is testable here.  A
    launcher/--gpus mismatch and a box with too few GPUs are errors, never a relabelled number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["comm"]["world_size"] == 2 and d["steps"] == 3 and d["allreduce_sum_check"] == 6.0
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub"], capture_output=True, text=True,
                       env={**env, "WORLD_SIZE": "1"}, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                           timeout=120)
        assert r.returncode != 0 and "refusing" in r.stderr


def test_bench_two_rank_train_path_end_to_end_on_cpu():
    """`python bench.py --gpus 2` through the TRAIN path itself, not the stub step: spawn, 127.0.0.1 rendezvous (gloo), parameter
    broadcast, the bucketed GradReducer inside fit_step, every rank's own clock gathered in timed(), the exchange summary, the
    efficiency field and the single rank-0 JSON line -- on a tiny model over the tests' CPU stand-ins of the kernels (--emu; the
    line's metric says it is not a measurement).  What cannot run here is RCCL; what can fail for a host-side reason on the
    first multi-GPU run is executed here (train.py:461-474)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "4"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--emu", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--seq", "16", "--baseline-1gpu", "1000", "--comm", "both"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["metric"].startswith("EMU") and d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["comm"]["world_size"] == 2 and d["comm"]["backend"] == "gloo" and d["comm"]["exchange"] == "torch"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 2 * 16 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]       # whole-job events / max-over-ranks time
    rk = d["ranks"]
    assert len(rk["ms_per_step_by_rank"]) == 2 and abs(rk["ms_per_step_max"] - d["ms_per_step"]) < 1e-9 and rk["straggler_ratio"] >= 1.0
    assert abs(d["scaling_efficiency"] - d["value"] / 2000.0) < 1e-12
    # one exchange of the whole flat gradient buffer per step, in several buckets
    assert d["allreduce_windows"] == 2 and d["allreduce_launches_per_optimizer_step"] >= 2
    assert d["allreduce_bytes_per_step"] > 4e6 and np.isfinite(d["loss"])


def test_bench_allreduce_summary_semantics():
    """bench.py's `allreduce_*` keys under --accumulate 2: the reducer reports one exchange per optimiser step (the whole flat
    gradient buffer, 467,685,376 B for tv2o-medium in bf16), i.e. one per TWO timed steps; per timed step that is half."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    flat = 467_685_376
    d = bench.summarize_allreduce([(flat, 15), (flat, 15), (flat, 15)], steps=6, exposed_ms=[0.5, 0.7, 0.6])
    assert d["allreduce_windows"] == 3 and d["allreduce_bytes_per_optimizer_step"] == flat
    assert d["allreduce_bytes_per_step"] == flat / 2 and d["allreduce_launches_per_optimizer_step"] == 15
    assert abs(d["allreduce_exposed_ms_per_optimizer_step"] - 0.6) < 1e-9 and abs(d["allreduce_exposed_ms_per_step"] - 0.3) < 1e-9
    d = bench.summarize_allreduce([(flat, 15)] * 5, steps=5)  # --accumulate 1 (the headline): one exchange per step
    assert d["allreduce_bytes_per_step"] == flat and d["allreduce_exposed_ms_per_step"] is None


def test_grad_reducer_refuses_an_untiled_buffer():
    """finish() checks that the ranges launched in a window tile the flat gradient buffer exactly once"""
    from midi_model_amd.train import GradReducer
    r = GradReducer(torch.zeros(100), None, bucket_bytes=1 << 20)
    r.world = 2                      # (no process group here: exercise the bookkeeping only)
    r._launch = lambda lo, hi: r.launched.append((lo, hi))
    r.ready(60, 100)
    r.ready(0, 50)                   # [50, 60) never announced
    with pytest.raises(RuntimeError, match="do not tile"):
        r.finish()
    # the failed window left nothing behind: the next, complete window goes through
    assert not r.launched and not r.works and r.pending is None
    r.ready(50, 100)
    r.ready(0, 50)
    r.finish()
    assert not r.launched and r.pending is None
    # a backward that legitimately skips ranges (frozen parameters) declares the ranges it covers
    r.expected = [(0, 40), (70, 100)]
    r.ready(70, 100)
    r.ready(0, 40)
    r.finish()
    r.ready(70, 100)
    with pytest.raises(RuntimeError, match="do not tile"):
        r.finish()
