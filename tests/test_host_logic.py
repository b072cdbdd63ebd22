"""Host-side logic on CPU: the kernel schedules (engine.py), the flat parameter layout, the autograd nodes,
the fused training step + optimiser, the KV-cached decode loop and generate(), all driven through the
test-only fake backend (tests/emu_ops.py) and checked against the reference-generated golden vectors.
The same checks run on the real HIP kernels in test_model_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

import midi_model_amd as mm
from midi_model_amd import engine
from midi_model_amd.train import TrainMIDIModel, lr_lambda

import emu_ops


def tiny_config():
    return mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512)


@pytest.fixture(scope="module")
def tok():
    return mm.MIDITokenizerV2()


@pytest.fixture()
def tiny(orc, tok):
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    batch = orc.synthetic_events(tok, 2, 17, seed=2)
    batch[1, 14:] = tok.pad_id
    return shp, sd, batch


def test_state_dict_surface(golden):
    g = golden("medium_forward.npz")
    model = mm.MIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium"))
    keys = list(model.state_dict().keys())
    assert set(keys) == set(str(k) for k in g["state_dict_keys"]) and len(keys) == 140
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"])
    # parameters are views of ONE flat buffer; q|k|v and gate|up are adjacent
    q, k = model.net.layers[0].self_attn.q_proj.weight, model.net.layers[0].self_attn.k_proj.weight
    assert k.data_ptr() == q.data_ptr() + q.numel() * q.element_size()
    g_, u = model.net.layers[3].mlp.gate_proj.weight, model.net.layers[3].mlp.up_proj.weight
    assert u.data_ptr() == g_.data_ptr() + g_.numel() * g_.element_size()
    assert model.net.embed_tokens.weight[0].abs().max() == 0  # pad row
    # .to() keeps the layout, load_state_dict writes through
    model = model.to(torch.bfloat16)
    assert model.dtype == torch.bfloat16 and model.lm_head.weight.dtype == torch.bfloat16
    w = model.net_token.layers[2].mlp.down_proj.weight
    assert model._flat.data_ptr() <= w.data_ptr() < model._flat.data_ptr() + model._flat.numel() * 2


def test_no_cpu_compute_path():
    model = mm.MIDIModel(tiny_config())
    with pytest.raises(RuntimeError, match="ROCm device"):
        model.forward(torch.zeros((1, 2, 8), dtype=torch.long))
    with pytest.raises(ValueError):
        with emu_ops.install():
            model.generate(np.zeros((3, 2, 8), dtype=np.int64), batch_size=2, max_len=4)


def test_api_path_forward_backward(orc, tiny, golden):
    shp, sd, batch = tiny
    g = golden("tiny_train.npz")
    with emu_ops.install():
        model = mm.MIDIModel(tiny_config())
        model.load_state_dict(sd, strict=True)
        x, y = batch[:, :-1].contiguous(), batch[:, 1:].contiguous()
        hidden = model.forward(x)
        np.testing.assert_allclose(hidden.detach().numpy(), g["hidden"], rtol=1e-4, atol=3e-5)
        h2 = hidden.reshape(-1, hidden.shape[-1])
        y2 = y.reshape(-1, y.shape[-1])
        logits = model.forward_token(h2, y2[:, :-1])
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, model.tokenizer.vocab_size), y2.reshape(-1),
                                                 reduction="mean", ignore_index=model.tokenizer.pad_id)
        loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 3e-5
    np.testing.assert_allclose(logits.detach()[:, :, ::16].numpy(), g["logits_sub"], rtol=1e-3, atol=3e-5)
    named = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([named[n].grad.norm().item() for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-4, atol=1e-7)


def test_cached_forward_matches_reference(tiny, golden):
    shp, sd, batch = tiny
    g = golden("tiny_train.npz")

    class AnyCache:  # forward() accepts whatever cache object the caller made (HF DynamicCache in app.py)
        pass

    with emu_ops.install(), torch.no_grad():
        model = mm.MIDIModel(tiny_config())
        model.load_state_dict(sd)
        x = batch[:, :-1]
        c = AnyCache()
        ha = model.forward(x[:, :11], cache=c)
        hb = model.forward(x[:, 11:], cache=c)
        h = torch.cat([ha, hb], 1)
    np.testing.assert_allclose(h.numpy(), g["hidden_cached"], rtol=1e-4, atol=3e-5)


def test_fused_step_grads_and_optimizer(orc, tiny, golden, tok):
    shp, sd, batch = tiny
    g = golden("tiny_train.npz")
    with emu_ops.install():
        model = TrainMIDIModel(tiny_config(), lr=1e-2, warmup=2, max_step=10, accumulate_grad_batches=1)
        model.load_state_dict(sd)
        loss = model.training_step(batch)
        assert abs(loss.item() - float(g["loss"])) < 3e-5
        named = dict(model.named_parameters())
        names = [str(n) for n in g["grad_names"]]
        norms = np.array([named[n].grad.norm().item() for n in names])
        np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-4, atol=1e-7)
        for key in g.files:
            if key.startswith("grad:"):
                gr = named[key[5:]].grad
                got = gr.numpy() if gr.dim() == 1 else gr[:64:3, ::5].numpy()
                np.testing.assert_allclose(got, g[key], rtol=2e-3, atol=2e-7)
        vloss, acc = model.validation_step(batch)
        assert abs(vloss.item() - float(g["loss"])) < 3e-5 and abs(acc.item() - float(g["acc"])) < 1e-6

        # three optimiser steps against the reference's torch.optim.AdamW + clip_grad_norm_ + LambdaLR
        model = TrainMIDIModel(tiny_config(), lr=1e-2, warmup=2, max_step=10, accumulate_grad_batches=1)
        model.load_state_dict(sd)
        losses, gn, lrs = [], [], []
        for step in range(3):
            b = orc.synthetic_events(tok, 2, 17, seed=10 + step)
            lrs.append(model.current_lr())
            losses.append(model.fit_step(b).item())
            gn.append(model.last_grad_norm.item())
        np.testing.assert_allclose(losses, g["opt_losses"], rtol=5e-5)
        np.testing.assert_allclose(gn, g["opt_gnorms"], rtol=5e-4)
        np.testing.assert_allclose(lrs, g["opt_lrs"], rtol=1e-12)
        named = dict(model.named_parameters())
        pn = np.array([named[n].detach().norm().item() for n in names])
        np.testing.assert_allclose(pn, g["opt_param_norms"], rtol=2e-5)
        for key in g.files:
            if key.startswith("opt:"):
                p = named[key[4:]].detach()
                got = p.numpy() if p.dim() == 1 else p[:64:3, ::5].numpy()
                np.testing.assert_allclose(got, g[key], rtol=2e-4, atol=2e-6)


def test_training_state_resume(orc, tiny, tok, tmp_path):
    """``trainer.fit(..., ckpt_path=opt.resume)`` (train.py:475-479): (a) save -> load into a fresh model -> continue equals the
    uninterrupted run bit for bit, also from the middle of an accumulation window; (b) a checkpoint in Lightning's layout
    written from the REAL ``torch.optim.AdamW`` / ``LambdaLR`` of train.py:121-151 after two steps of the reference recipe
    resumes here, and our third step lands where torch's third step lands."""
    shp, sd, _ = tiny
    batches = [orc.synthetic_events(tok, 2, 17, seed=40 + i) for i in range(4)]
    kw = dict(lr=1e-2, warmup=2, max_step=10)
    with emu_ops.install():
        for nacc, n_before in ((1, 2), (2, 3)):  # (2, 3): one optimiser step + one micro-batch into the next window
            a = TrainMIDIModel(tiny_config(), accumulate_grad_batches=nacc, **kw)
            a.load_state_dict(sd)
            for i in range(n_before):
                a.fit_step(batches[i])
            path = str(tmp_path / f"state_{nacc}.ckpt")
            a.save_training_state(path)
            b = TrainMIDIModel(tiny_config(), accumulate_grad_batches=nacc, **kw)
            b.load_training_state(path)
            assert b.global_step == a.global_step and b._micro == a._micro == n_before % nacc
            assert torch.equal(b._flat, a._flat) and torch.equal(b._opt["m"], a._opt["m"]) and torch.equal(b._opt["v"], a._opt["v"])
            la, lb = a.fit_step(batches[n_before]), b.fit_step(batches[n_before])
            assert torch.equal(la, lb) and torch.equal(b._flat, a._flat) and b.global_step == a.global_step
            assert torch.equal(b._opt["m"], a._opt["m"]) and torch.equal(b._opt["v"], a._opt["v"])
        with pytest.raises(RuntimeError, match="not a training checkpoint"):
            b.load_training_state({"state_dict": {}})

    # (b) the reference recipe with the real torch objects
    from midi_model_amd.train import lr_lambda
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    named = list(params.items())
    no_decay = ["bias", "norm"]
    opt = torch.optim.AdamW([{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
                             {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}],
                            lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s_: lr_lambda(s_, 2, 10))

    def ref_step(bt):
        opt.zero_grad()
        loss, _ = orc.training_loss(params, shp, bt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()
        sched.step()
        return loss.item()

    for i in range(2):
        ref_step(batches[i])
    ckpt = {"state_dict": {k: v.detach().clone() for k, v in params.items()}, "global_step": 2,
            "optimizer_states": [opt.state_dict()], "lr_schedulers": [sched.state_dict()]}
    path = str(tmp_path / "lightning_layout.ckpt")
    torch.save(ckpt, path)
    want_loss = ref_step(batches[2])
    with emu_ops.install():
        m = TrainMIDIModel(tiny_config(), accumulate_grad_batches=1, weight_decay=0.01, **kw)
        m.load_training_state(path)
        assert m.global_step == 2
        got_loss = m.fit_step(batches[2]).item()
    assert abs(got_loss - want_loss) < 5e-5 * abs(want_loss)
    for n, p in m.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), params[n].detach().numpy(), rtol=2e-4, atol=2e-6, err_msg=n)
    st = opt.state_dict()["state"]
    order = m._optimizer_param_order([n for n, _ in named])
    for i, n in enumerate(order):
        off, cnt, _ = m._offsets[n]
        np.testing.assert_allclose(m._opt["m"][off:off + cnt].numpy(), st[i]["exp_avg"].reshape(-1).numpy(), rtol=2e-3, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(m._opt["v"][off:off + cnt].numpy(), st[i]["exp_avg_sq"].reshape(-1).numpy(), rtol=2e-3, atol=1e-9, err_msg=n)

    # (c) the OTHER direction: what training_state() writes loads into the REAL torch.optim.AdamW and LambdaLR of the
    # reference recipe (LambdaLR.load_state_dict pops "lr_lambdas"; Lightning reads "pytorch-lightning_version"; "loops" is left out on purpose -- an empty dict would be indexed), and
    # torch's next step from there lands where ours lands
    state = m.training_state()
    assert {"pytorch-lightning_version", "state_dict", "optimizer_states", "lr_schedulers"} <= set(state) and "loops" not in state
    params2 = {k: state["state_dict"][k].clone().float().requires_grad_(True) for k, _ in named}
    named2 = list(params2.items())
    opt2 = torch.optim.AdamW([{"params": [p for n, p in named2 if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
                              {"params": [p for n, p in named2 if any(nd in n for nd in no_decay)], "weight_decay": 0.0}],
                             lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    sched2 = torch.optim.lr_scheduler.LambdaLR(opt2, lambda s_: lr_lambda(s_, 2, 10))
    opt2.load_state_dict(state["optimizer_states"][0])
    sched2.load_state_dict(state["lr_schedulers"][0])
    assert sched2.last_epoch == 3 and abs(sched2.get_last_lr()[0] - m.current_lr()) < 1e-12
    opt2.zero_grad()
    loss2, _ = orc.training_loss(params2, shp, batches[3])
    loss2.backward()
    torch.nn.utils.clip_grad_norm_(list(params2.values()), 1.0)
    opt2.step()
    with emu_ops.install():
        l3 = m.fit_step(batches[3]).item()
    assert abs(l3 - loss2.item()) < 5e-5 * abs(l3)
    for n, p in m.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), params2[n].detach().numpy(), rtol=2e-4, atol=1e-5, err_msg=n)
    # a checkpoint with objects beyond tensors needs the caller's word, per call
    import fractions
    torch.save(dict(ckpt, callbacks={"x": fractions.Fraction(1, 3)}), str(tmp_path / "odd.ckpt"))  # (not on torch's allow-list)
    with emu_ops.install():
        m2 = TrainMIDIModel(tiny_config(), accumulate_grad_batches=1, weight_decay=0.01, **kw)
        with pytest.raises(RuntimeError, match="trust_checkpoint"):
            m2.load_training_state(str(tmp_path / "odd.ckpt"))


def test_midimodel_is_a_mixin_base_the_way_the_reference_trainer_uses_it(orc, tiny, tok, golden):
    """``class TrainMIDIModel(MIDIModel, pl.LightningModule)`` (train.py:106-119): MIDIModel must cooperate as the FIRST of two
    module bases.  Lightning is not installed, so the second base is a stand-in with LightningModule's construction contract
    (an nn.Module subclass whose __init__ takes no arguments, sets its own attributes and offers log / hooks); the subclass is
    written as the reference writes it -- ``super(TrainMIDIModel, self).__init__(config)`` then its own fields -- and its
    training_step is the reference's (train.py:168-188) op for op on the public forward / forward_token surface."""
    import torch.nn.functional as F
    shp, sd, batch = tiny
    g = golden("tiny_train.npz")

    class StubLightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._logged = {}
            self.trainer = None

        def log(self, name, value, **kw):
            self._logged[name] = float(value.detach() if torch.is_tensor(value) else value)

        def on_train_start(self):
            return "hook"

    class RefStyleTrainModel(mm.MIDIModel, StubLightningModule):
        def __init__(self, config, lr=2e-4):
            super(RefStyleTrainModel, self).__init__(config)
            self.lr = lr
            self.last_save_step = 0

        def training_step(self, batch, batch_idx=0):  # train.py:168-188
            x = batch[:, :-1].contiguous()
            y = batch[:, 1:].contiguous()
            hidden = self.forward(x)
            hidden = hidden.reshape(-1, hidden.shape[-1])
            y = y.reshape(-1, y.shape[-1])
            x = y[:, :-1]
            logits = self.forward_token(hidden, x)
            loss = F.cross_entropy(logits.view(-1, self.tokenizer.vocab_size), y.view(-1), reduction="mean",
                                   ignore_index=self.tokenizer.pad_id)
            self.log("train/loss", loss)
            return loss

    assert [c.__name__ for c in RefStyleTrainModel.__mro__[:4]] == ["RefStyleTrainModel", "MIDIModel", "StubLightningModule", "Module"]
    with emu_ops.install():
        model = RefStyleTrainModel(tiny_config(), lr=1e-3)
        assert model._logged == {} and model.on_train_start() == "hook" and model.lr == 1e-3   # both bases initialised
        assert len(model.state_dict()) == len(sd) and isinstance(model, torch.nn.Module)
        model.load_state_dict(sd, strict=True)
        assert model.requires_grad_(True) is model and model.eval() is model and model.train() is model
        loss = model.training_step(batch)
        loss.backward()  # autograd through the drop-in's forward / forward_token (autograd.py)
        assert abs(loss.item() - float(g["loss"])) < 3e-5 and abs(model._logged["train/loss"] - float(g["loss"])) < 3e-5
        named = dict(model.named_parameters())
        names = [str(n) for n in g["grad_names"]]
        norms = np.array([named[n].grad.norm().item() for n in names])
        np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-4, atol=1e-7)


def test_grad_accumulation_averages_micro_batches(orc, tiny, tok):
    """accumulate_grad_batches=2 (the reference default, train.py:355): Lightning divides every micro-batch loss by the
    window length before its backward, so the window's gradient is the MEAN of the two micro-batch gradients (each
    micro-batch loss being its own token mean), and only then the optimiser runs."""
    shp, sd, _ = tiny
    b = orc.synthetic_events(tok, 4, 9, seed=21)
    with emu_ops.install():
        singles = []
        for half in (b[:2], b[2:]):
            m1 = TrainMIDIModel(tiny_config(), accumulate_grad_batches=1)
            m1.load_state_dict(sd)
            m1.training_step(half)
            singles.append(m1.grad_buffer().clone())
        m2 = TrainMIDIModel(tiny_config(), accumulate_grad_batches=2, lr=1e-2, warmup=0)
        m2.load_state_dict(sd)
        before = m2._flat.clone()
        m2.fit_step(b[:2])
        assert torch.equal(m2._flat, before) and m2.global_step == 0  # no optimiser step inside the window
        m2.training_step(b[2:])
        g2 = m2.grad_buffer().clone()
        m2.optimizer_step()
        assert m2.global_step == 1 and not torch.equal(m2._flat, before)
    np.testing.assert_allclose(g2.numpy(), (0.5 * (singles[0] + singles[1])).numpy(), rtol=1e-4, atol=1e-7)


def test_folded_norm_forward_only_stack(orc, tiny, tok, monkeypatch):
    """engine.stack_forward(save=False) with the RMSNorms folded around the projections (layer_forward_folded: statistics out of the
    o / down projections, rstd applied by the q|k|v and gate|up projections) against the plain blocks: the same hidden states
    within bf16 rounding (the fold moves two roundings: W' = round(w W) instead of round(w round(x rstd))), through the model's
    own no-grad forward, a cached prefill, and with a session-style pre-folded weight list.  Host schedule on the CPU stand-ins;
    the kernels are compared on the device (tests/test_kernels_gpu.py, test_parity_long_gpu.py)."""
    from midi_model_amd import engine
    shp, sd, batch = tiny
    x = orc.synthetic_events(tok, 2, 24, seed=8)
    with emu_ops.install():
        m = mm.MIDIModel(tiny_config())
        m.load_state_dict(sd)
        m = m.to(torch.bfloat16)
        with torch.no_grad():
            plain = m.forward(x).float()
        calls = []
        real = engine.layer_forward_folded
        monkeypatch.setattr(engine, "layer_forward_folded", lambda *a, **k: (calls.append(a[7] is not None), real(*a, **k))[1])
        monkeypatch.setattr(engine, "FOLD_MIN_ROWS_ON_THE_FLY", 0)
        monkeypatch.setattr(engine, "FOLD_MIN_ROWS_PREFOLDED", 0)
        with torch.no_grad():
            folded = m.forward(x).float()

            class AnyCache:
                pass
            cached = m.forward(x, cache=AnyCache()).float()
        # 4 layers x 2 calls; only the first block of a stack computes its statistics from the rows themselves
        assert calls == [False, True, True, True] * 2
    err = (folded - plain).abs().max().item()
    assert err < 0.06 * plain.abs().max().item(), err      # bf16 rounding noise through 4 layers, not a wiring error
    assert torch.equal(cached, folded)
    hid_o = orc.midi_forward(sd, shp, x)
    assert (folded - hid_o).abs().max() < 1.5 * (plain - hid_o).abs().max() + 0.02


def test_lean_activation_saving_is_bit_identical(orc, tiny, tok):
    """``lean_activations``: the forward drops the SwiGLU activations and the backward recomputes them from gate|up
    (engine.layer_forward / stack_backward) -- same loss, same gradients, bit for bit (host schedule; the device kernels share
    their roundings by construction, tests/test_model_gpu.py covers them)."""
    shp, sd, batch = tiny
    with emu_ops.install():
        outs = []
        for lean in (False, True):
            m = TrainMIDIModel(tiny_config(), accumulate_grad_batches=1)
            m.load_state_dict(sd)
            m.lean_activations = lean
            loss = m.training_step(batch)
            outs.append((loss.clone(), m.grad_buffer().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_training_fold_of_the_norms_is_the_same_gradient(orc, tiny, tok):
    """The training step with the RMSNorms folded around the projections (engine.layer_forward_train_folded /
    layer_backward_folded, r06) against the plain schedule on the same bf16 weights, through the stand-ins: the algebra of the
    fold -- d z = rstd (.) d y from the producers, t = d z W', dx = t - x (rstd^2 / D) rowdot(t, x) + dres, dW = (d z^T x) (.) w,
    dw = colsum((d z^T x) (.) W) -- must give the loss and EVERY gradient tensor of the unfolded graph up to bf16 rounding, the
    norm weights' gradients included (norm weights drawn away from one so that the fold is not trivial), also inside an
    accumulation window (second micro-batch added to the first)."""
    shp, sd, batch = tiny
    g = torch.Generator().manual_seed(7)
    sd = {k: (v * (1.0 + 0.3 * torch.randn(v.shape, generator=g)) if "norm" in k else v) for k, v in sd.items()}
    with emu_ops.install():
        outs = []
        for fold in (False, True):
            m = TrainMIDIModel(tiny_config(), accumulate_grad_batches=2)
            m.load_state_dict(sd)
            m = m.to(torch.bfloat16)
            m.fold_train_norms = fold
            calls = []
            real = engine.layer_backward_folded
            engine.layer_backward_folded = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
            try:
                l1 = m.training_step(batch)
                l2 = m.training_step(batch.flip(0))
            finally:
                engine.layer_backward_folded = real
            assert (len(calls) > 0) == fold, "the fold did not take the path it was asked for"
            outs.append((l1.float().item(), l2.float().item(), {k: p.grad.float().clone() for k, p in m.named_parameters()}))
    (a1, a2, ga), (b1, b2, gb) = outs
    assert abs(a1 - b1) < 2e-2 and abs(a2 - b2) < 2e-2, (a1, b1, a2, b2)
    worst = 0.0
    for k in ga:
        na, nb = ga[k].norm().item(), gb[k].norm().item()
        if na < 1e-12:
            assert nb < 1e-6, k
            continue
        cos = (ga[k] * gb[k]).sum().item() / (na * nb)
        worst = max(worst, 1.0 - cos)
        assert cos > 0.995 and abs(nb / na - 1.0) < 0.05, (k, cos, nb / na)
    assert worst < 5e-3


def test_generate_on_trained_weights_matches_reference(trained, tok):
    """the host side of generate() (grammar masks, break rule, sampler, RNG consumption) on PEAKED distributions: the tiny model
    trained by tests/gen_golden_trained.py, the reference's own seeded ids (midi_model.py:152-165, 195-248)"""
    shp, sd, g = trained
    with emu_ops.install():
        from conftest import trained_config
        model = mm.MIDIModel(trained_config())
        model.load_state_dict(sd)
        out = model.generate(None, batch_size=4, max_len=40, generator=torch.Generator().manual_seed(4321))
        assert out.shape == g["sampled_b4"].shape and (out == g["sampled_b4"]).all()
        out = model.generate(None, batch_size=4, max_len=40, top_k=1, generator=torch.Generator().manual_seed(0))
        assert (out == g["greedy_b4"]).all()
        out = model.generate(g["prompt"], batch_size=3, max_len=36, temp=0.9, top_p=0.9, top_k=8,
                             generator=torch.Generator().manual_seed(99))
        assert (out == g["prompt_sampled_b3"]).all()


def test_generate_matches_reference(tiny, golden, tok):
    shp, sd, _ = tiny
    g = golden("tiny_generate.npz")
    with emu_ops.install():
        model = mm.MIDIModel(tiny_config())
        model.load_state_dict(sd)
        out = model.generate(None, batch_size=3, max_len=14, generator=torch.Generator().manual_seed(1234))
        assert out.shape == g["sampled_b3"].shape and (out == g["sampled_b3"]).all()
        out = model.generate(None, batch_size=2, max_len=14, top_k=1, generator=torch.Generator().manual_seed(0))
        assert (out == g["greedy_b2"]).all()
        out = model.generate(g["prompt"], batch_size=2, max_len=12, temp=0.9, top_p=0.9, top_k=8,
                             generator=torch.Generator().manual_seed(77))
        assert (out == g["prompt_b2"]).all()
        pr = torch.softmax(3.0 * torch.randn((4, 1, tok.vocab_size), generator=torch.Generator().manual_seed(3)), -1)
        s = model.sample_top_p_k(pr, 0.9, 12, generator=torch.Generator().manual_seed(9))
        assert (s.numpy() == g["sampler_out"]).all()
        # ban_eos: every row runs to max_len and never emits EOS as an event id
        out = model.generate(None, batch_size=2, max_len=10, generator=torch.Generator().manual_seed(5), ban_eos=True)
        assert out.shape == (2, 10, 8) and (out[:, 1:, 0] != tok.eos_id).all()
        # the serving form (app.py:27-120): one (B, 8) array per event, same stream as generate()
        evs = list(model.generate_stream(None, batch_size=3, max_len=14, generator=torch.Generator().manual_seed(1234)))
        assert all(e.shape == (3, 8) and e.dtype == np.int64 for e in evs)
        assert (np.stack(evs, 1) == g["sampled_b3"][:, 1:]).all()
        # mask options of the serving loop: no patch_change / control_change events, no banned channel ids
        banned = [tok.parameter_ids["channel"][c] for c in (0, 9)]
        out = model.generate(None, batch_size=4, max_len=24, generator=torch.Generator().manual_seed(11), ban_eos=True,
                             disable_patch_change=True, disable_control_change=True, disable_channels=[0, 9])
        assert out.shape == (4, 24, 8)
        assert not np.isin(out[:, 1:, 0], [tok.event_ids["patch_change"], tok.event_ids["control_change"]]).any()
        assert not np.isin(out, banned).any()
        # a closed stream returns its session to the pool
        it = model.generate_stream(None, batch_size=3, max_len=14)
        next(it)
        it.close()
        assert len(model._sessions.idle) >= 1


def test_forward_token_decode_path(tiny, orc):
    """the reference calling convention of the inner loop: hidden first, then one token id at a time"""
    shp, sd, batch = tiny

    class C:
        pass

    with emu_ops.install(), torch.no_grad():
        model = mm.MIDIModel(tiny_config())
        model.load_state_dict(sd)
        hid = torch.randn(3, 256, generator=torch.Generator().manual_seed(0))
        toks = batch[0, 1:4, :4].contiguous()  # (3, 4) ids
        c = C()
        outs = [model.forward_token(hid, None, cache=c)]
        for j in range(4):
            outs.append(model.forward_token(None, toks[:, j:j + 1], cache=c))
        got = torch.cat(outs, 1)
    want = orc.midi_forward_token(sd, shp, hid, toks)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-3, atol=3e-5)


def test_lr_schedule():
    assert lr_lambda(0, 1000, 1e6) == 0.0 and lr_lambda(500, 1000, 1e6) == 0.5
    assert lr_lambda(1000, 1000, 1e6) == 1.0 and lr_lambda(10 ** 6, 1000, 1e6) == 0.0


def test_device_corpus_batches_follow_reference_collate(tok):
    """TokenCorpus + WindowSampler against MidiDataset.__getitem__ / collate_fn (train.py:69-90) restated on the host:
    same windows (same `random` stream), widened to int64, padded to the longest with pad_id"""
    import random
    from midi_model_amd.data import TokenCorpus, WindowSampler, synthetic_events
    lens = [50, 9, 300, 128, 64]
    pieces = [synthetic_events(tok, 1, n, seed=100 + i)[0].numpy().astype(np.int16) for i, n in enumerate(lens)]
    with emu_ops.install():
        corpus = TokenCorpus(pieces, device="cpu")
        for rand_start in (True, False):
            sampler = WindowSampler(corpus, max_len=64, rand_start=rand_start, seed=7)
            idx = [2, 0, 1, 4, 3, 2]
            got = sampler.batch(idx, pad_id=tok.pad_id)
            rng = random.Random(7)
            ref = []
            for i in idx:  # train.py:73-83
                mid = pieces[i]
                if rand_start:
                    start = rng.randrange(0, max(1, mid.shape[0] - 64))
                    start = rng.choice([0, start])
                else:
                    max_start = max(1, mid.shape[0] - 64)
                    start = (i * (max_start // 8)) % max_start
                ref.append(torch.from_numpy(mid[start:start + 64].astype(np.int64)))
            L = max(len(m) for m in ref)  # train.py:84-90
            want = torch.stack([torch.nn.functional.pad(m, (0, 0, 0, L - m.shape[0]), value=tok.pad_id) for m in ref])
            assert got.dtype == torch.int64 and torch.equal(got, want)


@pytest.mark.parametrize("ver", ["v1", "v2"])
def test_augment_decomposition_and_sampler_follow_the_reference(orc, golden, ver):
    """The device path splits MIDITokenizer.augment into per-file facts (computed once per corpus) + row-local rules applied
    while a window is cut (csrc/augment.hip).  Here that decomposition (the backend-shaped restatement in emu_ops) is held to
    the reference's own outputs on whole files, and WindowSampler(aug=True) to ``MidiDataset.load_midi`` + ``__getitem__`` +
    ``collate_fn`` (train.py:48-90) restated with the oracle: same `random` stream -- six shifts, then the window draws, per
    served file -- same windows, same augmented tokens."""
    import random
    from midi_model_amd.data import AUG_MAXIMA, TokenCorpus, WindowSampler
    from midi_model_amd.tokenizer import AUG_STATS, augment_table
    tok = mm.MIDITokenizerV1() if ver == "v1" else mm.MIDITokenizerV2()
    g = golden(f"augment_{ver}.npz")
    off = g["offsets"]
    P = len(off) - 1
    pieces = [g["tokens"][off[i]:off[i + 1]] for i in range(P)]
    tab = torch.tensor(augment_table(tok), dtype=torch.int32)
    tokens = torch.from_numpy(g["tokens"])
    stats = emu_ops.augment_piece_stats(tokens, torch.from_numpy(off), tab, torch.zeros((P, AUG_STATS), dtype=torch.int32))
    lens = torch.from_numpy(np.diff(off))
    out = torch.empty((P, int(lens.max()), tok.max_token_seq), dtype=torch.int64)
    emu_ops.augment_collate_windows(tokens, torch.from_numpy(off[:-1].copy()), lens, torch.arange(P), torch.from_numpy(g["shifts"]),
                                    stats, tab, out, tok.pad_id)
    for i in range(P):
        n = int(lens[i])
        assert np.array_equal(out[i, :n].numpy(), g["augmented"][off[i]:off[i + 1]].astype(np.int64)), (ver, i)
        assert (out[i, n:] == tok.pad_id).all()
    with emu_ops.install():
        corpus = TokenCorpus(pieces, device="cpu")
        sampler = WindowSampler(corpus, max_len=96, rand_start=True, seed=11, aug=True, tokenizer=tok)
        idx = [3, 0, 7, 7, 1, 10, 4, 2]
        got = sampler.batch(idx, pad_id=tok.pad_id)
    rng = random.Random(11)
    ref = []
    for i in idx:
        m = AUG_MAXIMA
        sh = [rng.randint(-m[0], m[0]), rng.randint(-m[1], m[1]), rng.randint(-m[2], m[2]), rng.randint(-m[3], m[3]),
              rng.randint(0, m[4]), rng.randint(0, m[5])]                       # tokenizer.augment's draws (midi_tokenizer.py:1025-1030)
        mid = orc.augment(tok, pieces[i], sh)                                   # load_midi (train.py:62-63)
        start = rng.randrange(0, max(1, mid.shape[0] - 96))                     # __getitem__ (train.py:73-78)
        start = rng.choice([0, start])
        ref.append(torch.from_numpy(mid[start:start + 96].astype(np.int64)))
    L = max(len(x) for x in ref)
    want = torch.stack([torch.nn.functional.pad(x, (0, 0, 0, L - x.shape[0]), value=tok.pad_id) for x in ref])
    assert got.dtype == torch.int64 and torch.equal(got, want)


def test_load_merge_lora(tmp_path, tiny):
    """a saved LoRA adapter merges as W += (alpha / r) * B @ A into exactly the targeted weights (midi_model.py:109-114)"""
    from safetensors.torch import save_file
    shp, sd, _ = tiny
    with emu_ops.install():
        model = mm.MIDIModel(tiny_config())
        model.load_state_dict(sd)
        before = {k: v.clone() for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(0)
        targets = ["net.layers.0.self_attn.q_proj", "net.layers.1.mlp.down_proj", "net_token.layers.0.mlp.gate_proj"]
        ad, want = {}, {}
        for t in targets:
            w = before[t + ".weight"]
            a, b = torch.randn((4, w.shape[1]), generator=g) * 0.1, torch.randn((w.shape[0], 4), generator=g) * 0.1
            ad[f"base_model.model.{t}.lora_A.weight"], ad[f"base_model.model.{t}.lora_B.weight"] = a, b
            want[t + ".weight"] = w + (8.0 / 4) * (b @ a)
        d = tmp_path / "adapter"
        d.mkdir()
        save_file(ad, str(d / "adapter_model.safetensors"))
        (d / "adapter_config.json").write_text(json.dumps({"peft_type": "LORA", "r": 4, "lora_alpha": 8,
                                                           "target_modules": ["q_proj", "down_proj", "gate_proj"]}))
        assert model.load_merge_lora(str(d)) is model
        after = model.state_dict()
        for k, v in after.items():
            ref = want.get(k, before[k])
            assert torch.allclose(v, ref, atol=1e-6), k
        with pytest.raises(FileNotFoundError):
            model.load_merge_lora("skytnt/some-hub-id")


def test_lora_training_step_matches_autograd(orc, tiny, tok, tmp_path):
    """train.py:439-449 (task lora): frozen base + rank-r adapters.  The fused step's adapter gradients equal autograd
    through the oracle on W + (alpha / r) B A; clip + AdamW touch the adapters only; the saved adapter merges back
    (midi_model.py:109-114) into the same effective weights."""
    shp, sd, batch = tiny
    r, alpha = 8, 16.0
    s = alpha / r
    with emu_ops.install():
        model = TrainMIDIModel(tiny_config(), lr=1e-2, warmup=0, max_step=100, accumulate_grad_batches=1, weight_decay=0.01)
        model.load_state_dict(sd)
        lo = model.add_adapter({"r": r, "lora_alpha": alpha}, generator=torch.Generator().manual_seed(5))
        assert len(lo.targets) == 7 * (4 + 1) and not any(p.requires_grad for p in model.parameters())
        g = torch.Generator().manual_seed(6)
        for name in lo.B:  # B = 0 would leave dA = 0: give the adapters something to do
            lo.B[name].copy_(torch.randn(lo.B[name].shape, generator=g) * 0.05)
        lo.dirty = True
        A0 = {k: v.clone() for k, v in lo.A.items()}
        B0 = {k: v.clone() for k, v in lo.B.items()}
        loss = model.training_step(batch)

        # autograd through the oracle on the effective weights
        A = {k: v.clone().requires_grad_(True) for k, v in A0.items()}
        B = {k: v.clone().requires_grad_(True) for k, v in B0.items()}
        sd_eff = dict(sd)
        for name in A:
            sd_eff[name + ".weight"] = sd[name + ".weight"] + s * (B[name] @ A[name])
        ref_loss, _ = orc.training_loss(sd_eff, shp, batch, tok.pad_id)
        ref_loss.backward()
        assert abs(loss.item() - ref_loss.item()) < 3e-5
        lo.compute_grads(model)
        for name in A:
            np.testing.assert_allclose(lo.gA[name].numpy(), A[name].grad.numpy(), rtol=2e-3, atol=2e-7, err_msg=name)
            np.testing.assert_allclose(lo.gB[name].numpy(), B[name].grad.numpy(), rtol=2e-3, atol=2e-7, err_msg=name)

        # optimiser step: clip_grad_norm_(1.0) + AdamW(lr, (0.9, 0.99), 1e-8, wd) over the adapter parameters only
        params = [p for d in (A, B) for p in d.values()]
        opt = torch.optim.AdamW(params, lr=model.current_lr(), betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        base_before = {k: v.clone() for k, v in model.state_dict().items()}
        model.optimizer_step()
        assert abs(model.last_grad_norm.item() - gn.item()) < 1e-3 * gn.item() and model.global_step == 1
        for name in A:  # (the first AdamW step is lr * g / (|g| + eps): elements with |g| ~ eps amplify fp32 noise)
            for got, want in ((lo.A[name], A[name].detach()), (lo.B[name], B[name].detach())):
                d = (got - want).abs()
                assert d.max().item() < 1e-2 * 0.1 and (d < 1e-6 + 1e-4 * want.abs()).float().mean().item() > 0.99, name

        # the next step runs on the updated effective weights; everything that is not a target stays as loaded
        loss2 = model.training_step(batch)
        after = model.state_dict()
        targets = {name + ".weight" for name in A}
        for k, v in after.items():
            if k in targets:
                want = sd[k] + s * (lo.B[k[:-7]] @ lo.A[k[:-7]])
                assert torch.allclose(v, want, atol=1e-6), k
            else:
                assert torch.equal(v, sd[k].to(v.dtype)), k
        assert loss2.item() < loss.item()

        # save -> merge into a fresh base model (midi_model.py:109-114) == the trained effective weights
        model.save_adapter(str(tmp_path / "adapter"))
        fresh = mm.MIDIModel(tiny_config())
        fresh.load_state_dict(sd)
        fresh.load_merge_lora(str(tmp_path / "adapter"))
        for k, v in fresh.state_dict().items():
            assert torch.allclose(v, after[k], atol=1e-6), k
        assert model.merge_and_unload() is model and model._lora is None
        with pytest.raises(NotImplementedError):
            model.add_adapter(r=8, lora_dropout=0.1)


def test_sample_seq_training_step(orc, tiny, tok):
    """train.py:172-175 (--sample-seq): the token-level stack and the loss see the last position + random others; the
    gradient of the sampled hidden states scatters back to their positions."""
    import random
    shp, sd, _ = tiny
    batch = orc.synthetic_events(tok, 2, 21, seed=31)
    with emu_ops.install():
        model = TrainMIDIModel(tiny_config(), sample_seq=True, accumulate_grad_batches=1)
        model.load_state_dict(sd)
        random.seed(1234)
        loss = model.training_step(batch)
    # the reference's lines, restated on the oracle with the same draw
    random.seed(1234)
    x, y = batch[:, :-1], batch[:, 1:]
    idx = [-1] + random.sample(list(range(y.shape[1] - 2)), min(127, (y.shape[1] - 2) // 2))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hidden = orc.midi_forward(sdg, shp, x)[:, idx].reshape(-1, shp.n_embd)
    ys = y[:, idx].reshape(-1, y.shape[-1])
    logits = orc.midi_forward_token(sdg, shp, hidden, ys[:, :-1])
    ref = torch.nn.functional.cross_entropy(logits.reshape(-1, shp.vocab), ys.reshape(-1), reduction="mean",
                                            ignore_index=tok.pad_id)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 3e-5
    named = dict(model.named_parameters())
    for k in ("net.embed_tokens.weight", "net.layers.0.self_attn.q_proj.weight", "net.layers.3.mlp.down_proj.weight",
              "net_token.layers.0.mlp.gate_proj.weight", "net.norm.weight", "lm_head.weight"):
        np.testing.assert_allclose(named[k].grad.numpy(), sdg[k].grad.numpy(), rtol=3e-3, atol=3e-7, err_msg=k)


def test_serving_loop_masks_and_crop_match_oracle_seeded(orc, tiny, tok):
    """generate() / generate_stream() with the serving loop's mask options (app.py:73-86) against the oracle's restatement
    of that loop, SEEDED (CPU generator, the reference's own draw): same ids.  Through the fake backend, so it runs here."""
    shp, sd, _ = tiny
    kw = dict(disable_patch_change=True, disable_control_change=True, disable_channels=[0, 9, 15])
    prompt = orc.synthetic_events(tok, 1, 4, seed=8)[0].numpy()
    with emu_ops.install():
        model = mm.MIDIModel(tiny_config())
        model.load_state_dict(sd)
        for seed, temp, top_p, top_k in ((21, 1.0, 0.98, 20), (22, 0.8, 0.9, 6)):
            want = orc.generate(sd, shp, tok, prompt, batch_size=3, max_len=20, temp=temp, top_p=top_p, top_k=top_k,
                                generator=torch.Generator().manual_seed(seed), **kw)
            got = model.generate(prompt, batch_size=3, max_len=20, temp=temp, top_p=top_p, top_k=top_k,
                                 generator=torch.Generator().manual_seed(seed), **kw)
            assert got.shape == want.shape and (got == want).all()
            evs = list(model.generate_stream(prompt, batch_size=3, max_len=20, temp=temp, top_p=top_p, top_k=top_k,
                                             generator=torch.Generator().manual_seed(seed), **kw))
            assert (np.stack(evs, 1) == want[:, 4:]).all()
        banned = [tok.parameter_ids["channel"][c] for c in (0, 9, 15)]
        assert not np.isin(want[:, 4:], banned).any()
        assert not np.isin(want[:, 4:, 0], [tok.event_ids["patch_change"], tok.event_ids["control_change"]]).any()
        with pytest.raises(ValueError, match="outside"):
            model.generate(np.full((2, 8), tok.vocab_size, dtype=np.int64), batch_size=1, max_len=4)
        # a mask emptied by the options is refused before any launch (the reference fails inside multinomial there)
        with pytest.raises(ValueError, match="every channel"):
            model.generate(prompt, batch_size=1, max_len=6, disable_channels=list(range(16)))
        with pytest.raises(ValueError, match="outside"):
            model.generate(prompt, batch_size=1, max_len=6, disable_channels=[16])


def test_reference_serving_loop_on_the_drop_in_with_real_dynamic_cache(orc, tiny, tok):
    """tests/ref_loops.py = app.py:27-120 verbatim.  It creates real transformers.DynamicCache objects and hands them to
    model.forward / model.forward_token, as app.py:56,64 does; with the same seeded generator its stream equals both
    generate()'s and the oracle's.  (GPU twin: tests/test_decode_gpu.py.)"""
    from transformers import DynamicCache
    from ref_loops import serving_loop
    import midi_model
    shp, sd, _ = tiny
    assert midi_model.MIDIModel is mm.MIDIModel and midi_model.config_name_list == mm.config_name_list

    class Mixin:
        pass

    class T(midi_model.MIDIModel, Mixin):  # train.py:106: class TrainMIDIModel(MIDIModel, pl.LightningModule)
        def __init__(self, config):
            super().__init__(config=config)

    with emu_ops.install():
        model = T(tiny_config())
        model.load_state_dict(sd, strict=False)
        model.eval()
        prompt = orc.synthetic_events(tok, 1, 5, seed=12)[0].numpy()
        evs = list(serving_loop(model, model.tokenizer, DynamicCache, prompt, batch_size=2, max_len=15,
                                disable_control_change=True, disable_channels=[9], generator=torch.Generator().manual_seed(4)))
        want = orc.generate(sd, shp, tok, prompt, batch_size=2, max_len=15, disable_control_change=True, disable_channels=[9],
                            generator=torch.Generator().manual_seed(4))
        assert (np.stack(evs, 1) == want[:, 5:]).all()
        got = model.generate(prompt, batch_size=2, max_len=15, disable_control_change=True, disable_channels=[9],
                             generator=torch.Generator().manual_seed(4))
        assert (got == want).all()


def test_checkpoint_forms_load_strictly(orc, tiny, tmp_path):
    """the on-disk forms either side of training -- save_pretrained dir, Lightning-shaped .ckpt, safetensors,
    prefixed keys -- load key-for-key into the flat buffer; benign extras are dropped, anything else raises."""
    from safetensors.torch import save_file
    shp, sd, _ = tiny
    model = mm.MIDIModel(tiny_config())
    model.load_state_dict(sd)
    model.save_pretrained(str(tmp_path / "hf"))
    again = mm.MIDIModel.from_pretrained(str(tmp_path / "hf"))
    assert again.config.to_dict() == model.config.to_dict()
    for k, v in again.state_dict().items():
        assert torch.equal(v, sd[k]), k
    state = {k: v.clone() for k, v in sd.items()}
    torch.save({"state_dict": state, "epoch": 1}, str(tmp_path / "a.ckpt"))
    torch.save({"state_dict": {"model." + k: v for k, v in state.items()}}, str(tmp_path / "b.ckpt"))
    save_file({**state, "net.rotary_emb.inv_freq": torch.ones(8)}, str(tmp_path / "c.safetensors"))
    for name in ("a.ckpt", "b.ckpt", "c.safetensors"):
        m = mm.MIDIModel.from_checkpoint(tiny_config(), str(tmp_path / name))
        assert torch.equal(m._flat, model._flat), name
    bad = dict(state)
    bad.pop("lm_head.weight")
    save_file(bad, str(tmp_path / "d.safetensors"))
    with pytest.raises(RuntimeError, match="missing"):
        mm.MIDIModel.from_checkpoint(tiny_config(), str(tmp_path / "d.safetensors"))
    save_file({**state, "net.layers.0.extra.weight": torch.ones(3)}, str(tmp_path / "e.safetensors"))
    with pytest.raises(RuntimeError, match="unexpected"):
        mm.MIDIModel.from_checkpoint(tiny_config(), str(tmp_path / "e.safetensors"))
    # wrapper prefixes in either nesting order (torch.compile inside DDP and the reverse)
    for i, pre in enumerate(("module._orig_mod.", "_orig_mod.module.", "model._orig_mod.module.")):
        save_file({pre + k: v for k, v in state.items()}, str(tmp_path / f"f{i}.safetensors"))
        m = mm.MIDIModel.from_checkpoint(tiny_config(), str(tmp_path / f"f{i}.safetensors"))
        assert torch.equal(m._flat, model._flat), pre
    # peft-wrapped keys are named for what they are
    wrapped = {k.replace("q_proj.weight", "q_proj.base_layer.weight"): v for k, v in state.items()}
    wrapped["net.layers.0.self_attn.q_proj.lora_A.default.weight"] = torch.ones(4, 4)
    save_file(wrapped, str(tmp_path / "g.safetensors"))
    with pytest.raises(RuntimeError, match="peft-wrapped"):
        mm.MIDIModel.from_checkpoint(tiny_config(), str(tmp_path / "g.safetensors"))
    # a pickle that is more than tensors: a clear refusal, and the reference's behaviour on request

    torch.save({"state_dict": state, "callbacks": {"x": _Opaque()}}, str(tmp_path / "h.ckpt"))
    with pytest.raises(RuntimeError, match="trust_checkpoint"):
        mm.MIDIModel.from_checkpoint(tiny_config(), str(tmp_path / "h.ckpt"))
    m = mm.MIDIModel.from_checkpoint(tiny_config(), str(tmp_path / "h.ckpt"), trust_checkpoint=True)
    assert torch.equal(m._flat, model._flat)


class _Opaque:  # (module level: picklable by reference, not on torch.load's allow-list)
    pass


def test_gen_example_and_save_peft_surface(tiny, golden, tmp_path):
    """TrainMIDIModel.gen_example (train.py:208-232, minus the Lightning / dataset globals): example_batch sequences from
    BOS and as many continuations of the prompt's first 256 events, saved per sequence; the same seeded stream as
    generate().  save_peft is the reference's name for save_adapter."""
    shp, sd, _ = tiny
    g = golden("tiny_generate.npz")
    with emu_ops.install():
        model = TrainMIDIModel(tiny_config(), example_batch=3)
        model.load_state_dict(sd)
        out = model.gen_example(str(tmp_path), max_len=14, generator=torch.Generator().manual_seed(1234))
        assert len(out) == 3 and all((o == w).all() for o, w in zip(out, g["sampled_b3"]))
        files = sorted(os.listdir(tmp_path / "sample" / "0"))
        assert files == ["0_0.npy", "0_1.npy", "0_2.npy"], files   # (tables-only tokenizer: no .mid / .png)
        assert (np.load(tmp_path / "sample" / "0" / "0_1.npy") == g["sampled_b3"][1]).all()
        out = model.gen_example(str(tmp_path / "p"), prompt=g["prompt"], max_len=12, generator=torch.Generator().manual_seed(3))
        assert len(out) == 6 and all((o[:g["prompt"].shape[-2]] == g["prompt"].reshape(-1, 8)[:o.shape[0]]).all() for o in out[3:])
        with pytest.raises(RuntimeError):
            model.save_peft(str(tmp_path / "lora"))   # no adapter attached
        model.add_adapter()
        model.save_peft(str(tmp_path / "lora"))
        assert sorted(os.listdir(tmp_path / "lora")) == ["adapter_config.json", "adapter_model.safetensors"]
