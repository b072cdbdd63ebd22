"""End-to-end parity on the MI355X against the reference-generated golden vectors (tests/golden, produced by
tests/gen_golden.py from /root/reference in fp32 on CPU):
  * fp32 mode: MIDIModel.forward / cached forward / forward_token / fused training step (loss, every gradient
    norm, gradient slices) / 3 optimiser steps / generate() token ids (seeded sampling, greedy, prompted) —
    tolerance rtol 1e-3 on logits (north_star), token ids bit-exact;
  * bf16 mode (production): error vs the fp32 reference bounded by the reference's OWN bf16-vs-fp32 drift
    (hidden 0.091, logits 0.054 max-abs at random init);
  * size-independent properties at the BASELINE shape (S=2048): finite loss near ln(vocab), run-to-run
    determinism of the loss, bf16 and fp32 gradients pointing the same way.
"""
import math

import numpy as np
import pytest
import torch

import midi_model_amd as mm
from midi_model_amd.train import TrainMIDIModel

pytestmark = pytest.mark.gpu


def tiny_config():
    return mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512)


@pytest.fixture(scope="module")
def tok():
    return mm.MIDITokenizerV2()


@pytest.fixture(scope="module")
def tiny(orc, tok):
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    batch = orc.synthetic_events(tok, 2, 17, seed=2)
    batch[1, 14:] = tok.pad_id
    return shp, sd, batch


def build(cls, cfg, sd, dtype=torch.float32, **kw):
    m = cls(cfg, **kw)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda", dtype)


def test_library_is_loaded_and_no_fallback():
    from midi_model_amd.lib import lib, LIB_PATH
    assert lib().cdll.mh_version() >= 1
    maps = open("/proc/self/maps").read()
    assert LIB_PATH in maps, "libmidihip.so is not mapped into the test process"


def test_tiny_fp32_forward_cache_and_api_backward(tiny, golden):
    shp, sd, batch = tiny
    g = golden("tiny_train.npz")
    model = build(mm.MIDIModel, tiny_config(), sd)
    x, y = batch[:, :-1].contiguous().cuda(), batch[:, 1:].contiguous().cuda()
    hidden = model.forward(x)
    np.testing.assert_allclose(hidden.detach().cpu().numpy(), g["hidden"], rtol=1e-3, atol=1e-4)

    class C:
        pass

    with torch.no_grad():
        c = C()
        h = torch.cat([model.forward(x[:, :11], cache=c), model.forward(x[:, 11:], cache=c)], 1)
    np.testing.assert_allclose(h.cpu().numpy(), g["hidden_cached"], rtol=1e-3, atol=1e-4)
    # reference-style step through the autograd nodes
    h2 = hidden.reshape(-1, hidden.shape[-1])
    y2 = y.reshape(-1, y.shape[-1])
    logits = model.forward_token(h2, y2[:, :-1])
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, model.tokenizer.vocab_size).float(), y2.reshape(-1),
                                             reduction="mean", ignore_index=model.tokenizer.pad_id)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    lg = logits.detach().cpu()
    np.testing.assert_allclose(lg[:, :, ::16].numpy(), g["logits_sub"], rtol=1e-3, atol=1e-4)
    assert (lg.argmax(-1).numpy() == g["logits_argmax"]).all()
    named = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([named[n].grad.norm().item() for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)


def test_tiny_fp32_fused_step_and_optimizer(orc, tiny, golden, tok):
    shp, sd, batch = tiny
    g = golden("tiny_train.npz")
    model = build(TrainMIDIModel, tiny_config(), sd, lr=1e-2, warmup=2, max_step=10, accumulate_grad_batches=1)
    loss = model.training_step(batch)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    named = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([named[n].grad.norm().item() for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for key in g.files:
        if key.startswith("grad:"):
            gr = named[key[5:]].grad.cpu()
            got = gr.numpy() if gr.dim() == 1 else gr[:64:3, ::5].numpy()
            np.testing.assert_allclose(got, g[key], rtol=5e-3, atol=1e-6)
    vloss, acc = model.validation_step(batch)
    assert abs(vloss.item() - float(g["loss"])) < 1e-4 and abs(acc.item() - float(g["acc"])) < 1e-6

    model = build(TrainMIDIModel, tiny_config(), sd, lr=1e-2, warmup=2, max_step=10, accumulate_grad_batches=1)
    losses, gn = [], []
    for step in range(3):
        b = orc.synthetic_events(tok, 2, 17, seed=10 + step)
        losses.append(model.fit_step(b).item())
        gn.append(model.last_grad_norm.item())
    np.testing.assert_allclose(losses, g["opt_losses"], rtol=2e-4)
    np.testing.assert_allclose(gn, g["opt_gnorms"], rtol=2e-3)
    named = dict(model.named_parameters())
    pn = np.array([named[n].detach().norm().item() for n in names])
    np.testing.assert_allclose(pn, g["opt_param_norms"], rtol=1e-4)
    for key in g.files:
        if key.startswith("opt:"):
            p = named[key[4:]].detach().cpu()
            got = p.numpy() if p.dim() == 1 else p[:64:3, ::5].numpy()
            np.testing.assert_allclose(got, g[key], rtol=1e-3, atol=1e-5)


def test_tiny_fp32_generate_token_ids(tiny, golden, tok):
    """generate() in fp32 reproduces the reference's token ids.  torch.multinomial consumes a CUDA generator
    differently from a CPU one, so the seeded-sampling cases draw on the CPU generator the reference used."""
    shp, sd, _ = tiny
    g = golden("tiny_generate.npz")
    model = build(mm.MIDIModel, tiny_config(), sd)
    out = model.generate(None, batch_size=2, max_len=14, top_k=1, generator=None)
    assert out.shape == g["greedy_b2"].shape and (out == g["greedy_b2"]).all()
    out = model.generate(None, batch_size=2, max_len=10, ban_eos=True)
    assert out.shape == (2, 10, 8) and (out[:, 1:, 0] != tok.eos_id).all()
    for b in range(2):
        for row in out[b, 1:]:
            assert tok.tokens2event(row.tolist()) != [], row  # every generated octet is a well-formed event
    with pytest.raises(ValueError):
        model.generate(np.zeros((3, 2, 8), dtype=np.int64), batch_size=2, max_len=4)
    # serving form (app.py:27-120): streamed events equal the batch call on the same seed; mask options hold
    gen = torch.Generator(device="cuda")
    ref = model.generate(None, batch_size=3, max_len=12, generator=gen.manual_seed(5))
    evs = list(model.generate_stream(None, batch_size=3, max_len=12, generator=gen.manual_seed(5)))
    assert (np.stack(evs, 1) == ref[:, 1:]).all()
    banned = [tok.parameter_ids["channel"][c] for c in (0, 9)]
    out = model.generate(None, batch_size=4, max_len=24, generator=gen.manual_seed(6), ban_eos=True,
                         disable_patch_change=True, disable_control_change=True, disable_channels=[0, 9])
    assert out.shape == (4, 24, 8)
    assert not np.isin(out[:, 1:, 0], [tok.event_ids["patch_change"], tok.event_ids["control_change"]]).any()
    assert not np.isin(out, banned).any()
    # top_k above the fused sampler's limit: the reference's own op chain (torch.sort / cumsum / multinomial) inside the graphs
    a = model.generate(None, batch_size=2, max_len=10, top_k=100, ban_eos=True, generator=gen.manual_seed(8))
    b = model.generate(None, batch_size=2, max_len=10, top_k=100, ban_eos=True, generator=gen.manual_seed(8))
    assert a.shape == (2, 10, 8) and (a == b).all()
    assert all(tok.tokens2event(r.tolist()) != [] for i in range(2) for r in a[i, 1:])
    # bf16 + graphs: same API, well-formed events
    mb = build(mm.MIDIModel, tiny_config(), sd, dtype=torch.bfloat16)
    out = mb.generate(None, batch_size=2, max_len=10, ban_eos=True, generator=gen.manual_seed(7))
    assert out.shape == (2, 10, 8) and all(tok.tokens2event(r.tolist()) != [] for b in range(2) for r in out[b, 1:])


def test_bf16_decode_session_variants_agree(tiny, tok, monkeypatch):
    """The bf16 decode step has three forms -- eager, captured graphs, graphs with the RMSNorm weights folded into the
    projections -- which differ only in where bf16 roundings fall: after a prompt prefill and one decoded event their
    hidden states and lm_head logits must agree within bf16 noise, and the eager and graph forms (same kernels) exactly."""
    from midi_model_amd.decode import DecodeSession
    shp, sd, batch = tiny
    model = build(mm.MIDIModel, tiny_config(), sd, dtype=torch.bfloat16)
    prompt = batch[:2, :5].cuda()

    def run(graphs: str, fold: str):
        monkeypatch.setenv("MH_DECODE_GRAPHS", graphs)
        monkeypatch.setenv("MH_DECODE_FOLD", fold)
        with torch.inference_mode():
            ses = DecodeSession(model, 2, 256, 1.0, 0.98, 1)
            ses.first_mask.copy_(model._grammar()[0])
            ses.reset()
            ses.begin(torch.Generator(device="cuda").manual_seed(3))
            ses.prefill(prompt)
            h0 = ses.hidden.float().clone()
            ses.tok_step(0)
            l0 = ses.logits[:, : tok.vocab_size].float().clone()
            ses.seq.copy_(batch[:2, 5].cuda())
            ses.consumed(1)  # the draws of the step above are reported before the next event's step 0 (decode.py contract)
            ses.net_step()
            h1 = ses.hidden.float().clone()
            ses.tok_step(0)
            l1 = ses.logits[:, : tok.vocab_size].float().clone()
            ses.end()
        return h0, l0, h1, l1

    eager = run("0", "0")
    graph = run("1", "0")
    folded = run("1", "1")
    for a, b in zip(eager, graph):
        assert torch.equal(a, b)
    for a, b, what in zip(graph, folded, ("prefill hidden", "logits 0", "decoded hidden", "logits 1")):
        err = (a - b).abs().max().item()
        assert err <= 0.03 * a.abs().max().item() + 1e-3, (what, err, a.abs().max().item())


def test_medium_fp32_matches_reference(orc, golden, tok):
    g = golden("medium_forward.npz")
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    model = build(TrainMIDIModel, mm.MIDIModelConfig.from_name("tv2o-medium"), sd)
    batch = orc.synthetic_events(tok, 1, 33, seed=4).cuda()
    with torch.no_grad():
        hidden = model.forward(batch[:, :-1])
        h2 = hidden.reshape(-1, 1024)
        y2 = batch[:, 1:].reshape(-1, 8)
        logits = model.forward_token(h2, y2[:, :-1]).float().cpu()
        loss, _ = model.validation_step(batch)
    assert abs(loss.item() - float(g["loss"])) < 2e-4
    np.testing.assert_allclose(h2[:, ::4].cpu().numpy(), g["hidden_sub"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(logits[:, :, ::32].numpy(), g["logits_sub"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(torch.logsumexp(logits, -1).numpy(), g["logits_lse"], rtol=1e-4, atol=1e-4)
    safe = g["logits_margin"] > 1e-3
    assert (logits.argmax(-1).numpy() == g["logits_argmax"])[safe].all()


def test_medium_bf16_within_reference_drift(orc, golden, tok):
    """Production dtype.  The reference in bf16 differs from its own fp32 by 0.091 (hidden) / 0.054 (logits)
    max-abs and 3.7 % of argmaxes at random init; we must be no worse than ~1.5x that."""
    g = golden("medium_forward.npz")
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    model = build(TrainMIDIModel, mm.MIDIModelConfig.from_name("tv2o-medium"), sd, dtype=torch.bfloat16)
    batch = orc.synthetic_events(tok, 1, 33, seed=4).cuda()
    with torch.no_grad():
        hidden = model.forward(batch[:, :-1])
        h2 = hidden.reshape(-1, 1024)
        y2 = batch[:, 1:].reshape(-1, 8)
        logits = model.forward_token(h2, y2[:, :-1]).float().cpu()
        loss, _ = model.validation_step(batch)
    assert abs(loss.item() - float(g["loss"])) < 3e-2
    herr = np.abs(h2[:, ::4].float().cpu().numpy() - g["hidden_sub"]).max()
    lerr = np.abs(logits[:, :, ::32].numpy() - g["logits_sub"]).max()
    agree = (logits.argmax(-1).numpy() == g["logits_argmax"]).mean()
    print(f"bf16 drift: hidden {herr:.4f} logits {lerr:.4f} argmax agreement {agree:.4f}")
    assert herr < 0.14 and lerr < 0.09 and agree > 0.93
    safe = g["logits_margin"] > 0.15
    assert (logits.argmax(-1).numpy() == g["logits_argmax"])[safe].all()


def test_baseline_shape_properties(orc, tok):
    """tv2o-medium, bf16, S=2048 events (BASELINE config 2 per sequence; B=2 to keep the test short)."""
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    model = build(TrainMIDIModel, mm.MIDIModelConfig.from_name("tv2o-medium"), sd, dtype=torch.bfloat16,
                  accumulate_grad_batches=1)
    batch = orc.synthetic_events(tok, 2, 2049, seed=6).cuda()
    l1 = model.training_step(batch).item()
    g1 = model.grad_buffer().float().clone()
    l2 = model.training_step(batch).item()
    assert math.isfinite(l1) and abs(l1 - math.log(tok.vocab_size)) < 1.0
    assert l1 == l2, "the loss must be reproducible run to run"
    assert torch.isfinite(g1).all() and g1.norm().item() > 0
    # the last event's hidden state does not depend on how the prefix was batched: causal consistency
    with torch.no_grad():
        full = model.forward(batch[:1, :512])
        part = model.forward(batch[:1, :300])
    d = (full[:, :300].float() - part.float()).abs().max().item()
    assert d < 0.05, d


def test_bf16_gradients_track_fp32(orc, tok):
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    cfg = mm.MIDIModelConfig.from_name("tv2o-medium")
    batch = orc.synthetic_events(tok, 2, 257, seed=8).cuda()
    gs = []
    for dtype in (torch.float32, torch.bfloat16):
        model = build(TrainMIDIModel, cfg, sd, dtype=dtype, accumulate_grad_batches=1)
        loss = model.training_step(batch)
        gs.append((loss.item(), model.grad_buffer().float().clone()))
        del model
    assert abs(gs[0][0] - gs[1][0]) < 3e-2
    cos = torch.nn.functional.cosine_similarity(gs[0][1], gs[1][1], dim=0).item()
    print(f"fp32 vs bf16 gradient cosine {cos:.5f}")
    assert cos > 0.98


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_lean_activation_saving_is_bit_identical_on_the_device(orc, tok, dtype):
    """``TrainMIDIModel.lean_activations`` (the forward drops the SwiGLU activations, the backward recomputes them from gate|up
    with mh_swiglu_fwd): loss and every gradient element equal the full-activation step's, bit for bit -- the fused gate|up
    epilogue and mh_swiglu_fwd share their roundings.  Shape with a fused-epilogue MLP (inner 1024 / 256), 4 x 128 events."""
    cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 1024)
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=1024, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=21)
    batch = orc.synthetic_events(tok, 4, 129, seed=22)
    outs = []
    for lean in (False, True):
        m = build(TrainMIDIModel, cfg, sd, dtype, accumulate_grad_batches=1)
        m.lean_activations = lean
        loss = m.training_step(batch)
        outs.append((loss.float().cpu().clone(), m.grad_buffer().float().cpu().clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    if dtype == torch.bfloat16:
        assert torch.equal(outs[0][1], outs[1][1])
    else:  # fp32: the embedding gradients are summed with fp32 atomics per run of equal ids (order varies launch to launch: the
        #       last bits of those rows differ between ANY two runs of the same step); everything else is exact
        torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-8)
        assert (outs[0][1] == outs[1][1]).float().mean() > 0.9
    assert torch.isfinite(outs[0][1]).all() and outs[0][1].abs().max() > 0


def test_device_corpus_feeds_training_step(tiny, tok):
    """the step before the path (data.TokenCorpus / WindowSampler, one-kernel batch assembly on the device) feeding the
    training step: ragged pieces -> padded (B, L, 8) int64 batch -> finite loss, pad targets ignored"""
    from midi_model_amd.data import TokenCorpus, WindowSampler, synthetic_events
    from midi_model_amd.train import TrainMIDIModel
    shp, sd, _ = tiny
    pieces = [synthetic_events(tok, 1, n, seed=200 + i)[0].numpy().astype(np.int16) for i, n in enumerate([40, 17, 64, 33])]
    corpus = TokenCorpus(pieces, device="cuda")
    sampler = WindowSampler(corpus, max_len=32, rand_start=True, seed=1)
    batch = sampler.batch([0, 1, 2, 3], pad_id=tok.pad_id)
    assert batch.is_cuda and batch.dtype == torch.int64 and batch.shape == (4, 32, 8)
    assert (batch[1, 17:] == tok.pad_id).all() and (batch[1, :17].cpu() == torch.from_numpy(pieces[1].astype(np.int64))).all()
    model = build(TrainMIDIModel, tiny_config(), sd, accumulate_grad_batches=1)
    loss = model.training_step(batch)
    assert torch.isfinite(loss).all()
    short = model.training_step(batch[1:2, :17])  # the unpadded piece alone
    both = model.training_step(torch.cat([batch[1:2], batch[1:2]], 0))  # the same piece twice, padded rows ignored
    assert abs(float(short) - float(both)) < 1e-4 * abs(float(short)) + 1e-5


def test_concurrent_generators_on_one_model(tiny, tok):
    """app.py runs up to 10 generators on one model (app.py:496): two threads on their own streams, each creating (and
    capturing) its own decode session at the same time, must both produce the stream a lone call produces"""
    import threading
    shp, sd, _ = tiny
    model = build(mm.MIDIModel, tiny_config(), sd, dtype=torch.bfloat16)
    want = [model.generate(None, batch_size=2 + i, max_len=12, ban_eos=True,
                           generator=torch.Generator(device="cuda").manual_seed(40 + i)) for i in range(2)]
    for rep in range(8):  # (r04: one thread's noise-graph replay inside the other's capture window raised under HIP about once
        #                      in three runs of the whole suite; several rounds of fresh captures make the window likely here)
        model._sessions.idle.clear()  # force both threads to build and capture new sessions
        got, errs = [None, None], []
        gens = [torch.Generator(device="cuda").manual_seed(40 + i) for i in range(2)]  # (seeded before any capture can be active)

        def work(i):
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    got[i] = model.generate(None, batch_size=2 + i, max_len=12, ban_eos=True, generator=gens[i])
                    torch.cuda.current_stream().synchronize()
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert not errs, (rep, errs)
        for i in range(2):
            assert (got[i] == want[i]).all(), rep


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lora_training_on_device(orc, tiny, tok, tmp_path, dtype):
    """train.py:439-449: adapters on a frozen base.  fp32: adapter gradients equal autograd through the oracle on
    W + (alpha / r) B A.  Both dtypes: a few optimiser steps lower the loss, nothing but the targeted weights moves,
    and the saved adapter merges (midi_model.py:109-114) into the trained effective weights."""
    shp, sd, batch = tiny
    r, alpha = 8, 16.0
    s = alpha / r
    model = build(TrainMIDIModel, tiny_config(), sd, dtype, lr=5e-3, warmup=0, max_step=100, accumulate_grad_batches=1)
    lo = model.add_adapter(r=r, lora_alpha=alpha, generator=torch.Generator().manual_seed(5))
    g = torch.Generator().manual_seed(6)
    for name in lo.B:
        lo.B[name].copy_((torch.randn(lo.B[name].shape, generator=g) * 0.05).to(lo.B[name]))
    lo.dirty = True
    A0 = {k: v.float().cpu() for k, v in lo.A.items()}
    B0 = {k: v.float().cpu() for k, v in lo.B.items()}
    loss = model.training_step(batch)
    if dtype == torch.float32:
        A = {k: v.clone().requires_grad_(True) for k, v in A0.items()}
        B = {k: v.clone().requires_grad_(True) for k, v in B0.items()}
        sd_eff = dict(sd)
        for name in A:
            sd_eff[name + ".weight"] = sd[name + ".weight"] + s * (B[name] @ A[name])
        ref_loss, _ = orc.training_loss(sd_eff, shp, batch, tok.pad_id)
        ref_loss.backward()
        assert abs(loss.item() - ref_loss.item()) < 1e-4
        lo.compute_grads(model)
        for name in A:
            np.testing.assert_allclose(lo.gA[name].cpu().numpy(), A[name].grad.numpy(), rtol=5e-3, atol=1e-6, err_msg=name)
            np.testing.assert_allclose(lo.gB[name].cpu().numpy(), B[name].grad.numpy(), rtol=5e-3, atol=1e-6, err_msg=name)
    losses = [loss.item()]
    model.optimizer_step()
    for _ in range(4):
        losses.append(model.fit_step(batch).item())
    assert losses[-1] < losses[0] - 0.01, losses
    model.training_step(batch)  # (materialises the latest adapters)
    after = {k: v.float().cpu() for k, v in model.state_dict().items()}
    targets = {name + ".weight" for name in lo.A}
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for k, v in after.items():
        ref = sd[k].to(dtype).float()
        if k in targets:
            ref = ref + s * (lo.B[k[:-7]].float().cpu() @ lo.A[k[:-7]].float().cpu())
            assert torch.allclose(v, ref, atol=tol), k
        else:
            assert torch.equal(v, ref), k
    model.save_adapter(str(tmp_path / "adapter"))
    fresh = build(mm.MIDIModel, tiny_config(), sd, dtype)
    fresh.load_merge_lora(str(tmp_path / "adapter"))
    for k, v in fresh.state_dict().items():
        assert torch.allclose(v.float().cpu(), after[k], atol=tol), k
