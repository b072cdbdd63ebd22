"""Shape fuzz of the public surface against the oracle (r06): random widths / depths / batch and sequence lengths -- ragged, odd,
one-event, one-sequence, rows not a multiple of any tile -- through MIDIModel.forward (grad and no_grad: the folded-norm path gates
on the row count), the cached forward split at a random point, forward_token, and the fused training step (loss + every gradient
norm).  fp32 against the oracle at north_star's rtol 1e-3; the bf16 forward inside a coarse range-relative bound (a wrong path at
an odd shape, not rounding, is what it guards).  Seeds are fixed: a failure names its case."""
import numpy as np
import pytest
import torch

import midi_model_amd as mm
from midi_model_amd.train import TrainMIDIModel

pytestmark = pytest.mark.gpu

# (n_layer, n_head, n_embd, n_inner, B, S): head_dim is 64 at the event level and 256 at the token level by construction
CASES = [
    (4, 4, 256, 512, 1, 1), (4, 4, 256, 512, 3, 2), (4, 4, 256, 768, 2, 63), (4, 4, 256, 512, 5, 65), (4, 8, 512, 1024, 1, 129),
    (8, 4, 256, 1024, 2, 31), (4, 8, 512, 1536, 3, 77), (4, 12, 768, 2048, 2, 50), (4, 4, 256, 512, 7, 37), (4, 8, 512, 1024, 4, 255),
    (4, 4, 256, 512, 1, 300), (4, 8, 512, 2048, 2, 257),
]


def _mk(orc, tok, case, seed):
    L, H, D, I, B, S = case
    shp = orc.Shape(n_layer=L, n_head=H, n_embd=D, n_inner=I, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=seed, std=0.05)
    g = torch.Generator().manual_seed(seed)
    for k in sd:  # norm weights away from 1 so that a dropped / doubled norm weight shows
        if k.endswith("norm.weight") or "layernorm" in k:
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    batch = orc.synthetic_events(tok, B, S + 1, seed=seed + 1)
    if S > 4 and B > 1:  # a ragged tail on one sequence (pad rows: no loss, no embedding gradient)
        batch[B - 1, S - 2:] = tok.pad_id
    cfg = mm.MIDIModelConfig.get_config("v2", True, L, H, D, I)
    return shp, sd, batch, cfg


@pytest.fixture(scope="module")
def tok():
    return mm.MIDITokenizerV2()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_fp32_surface_matches_oracle_on_fuzzed_shapes(orc, tok, ci):
    case = CASES[ci]
    shp, sd, batch, cfg = _mk(orc, tok, case, 100 + ci)
    B, S = case[4], case[5]
    x, y = batch[:, :-1].contiguous(), batch[:, 1:].contiguous()
    with torch.no_grad():
        ref_h = orc.midi_forward(sd, shp, x)
    model = mm.MIDIModel(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda", torch.float32)
    xc = x.cuda()
    h_grad = model.forward(xc)                       # saves for backward: the training forward
    with torch.no_grad():
        h_ng = model.forward(xc)                     # forward-only form
    tol = dict(rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(h_grad.detach().cpu().numpy(), ref_h.numpy(), **tol)
    np.testing.assert_allclose(h_ng.cpu().numpy(), ref_h.numpy(), **tol)
    if S >= 2:  # cached forward split at a case-dependent point

        class C:
            pass

        cut = 1 + (7 * ci) % (S - 1)
        with torch.no_grad():
            c = C()
            hc = torch.cat([model.forward(xc[:, :cut], cache=c), model.forward(xc[:, cut:], cache=c)], 1)
        np.testing.assert_allclose(hc.cpu().numpy(), ref_h.numpy(), **tol)
    # token level + loss through the autograd nodes, against the oracle's loss and gradient norms
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss, ref_logits = orc.training_loss(sdg, shp, batch, tok.pad_id)
    ref_loss.backward()
    with torch.no_grad():
        lg = model.forward_token(h_ng.reshape(-1, shp.n_embd), y.reshape(-1, 8)[:, :-1].cuda())
    np.testing.assert_allclose(lg.cpu().numpy(), ref_logits.detach().numpy(), rtol=1e-3, atol=5e-4)
    tm = TrainMIDIModel(cfg, lr=1e-3, warmup=2, max_step=10, accumulate_grad_batches=1)
    tm.load_state_dict(sd, strict=True)
    tm = tm.to("cuda", torch.float32)
    loss = tm.training_step(batch)
    assert abs(loss.item() - ref_loss.item()) < 2e-4 * max(1.0, abs(ref_loss.item())), (loss.item(), ref_loss.item())
    named = dict(tm.named_parameters())
    for n, p in sdg.items():
        gr = named[n].grad
        rn = p.grad.norm().item() if p.grad is not None else 0.0
        assert abs(gr.norm().item() - rn) <= 3e-3 * rn + 1e-6, (n, gr.norm().item(), rn)
    # one full gradient, element by element
    n0 = "net.layers.0.self_attn.q_proj.weight"
    np.testing.assert_allclose(named[n0].grad.cpu().numpy(), sdg[n0].grad.numpy(), rtol=2e-2,
                               atol=2e-3 * sdg[n0].grad.abs().max().item() + 1e-7)  # (S = 1: one key, d q is exactly 0 in the oracle)


@pytest.mark.parametrize("ci", [2, 4, 7, 9, 11])
def test_bf16_forward_tracks_the_oracle_on_fuzzed_shapes(orc, tok, ci):
    """bf16 (production dtype) on the same shapes: a coarse bound -- 4 % of the output's range, about twice the reference's own
    bf16-vs-fp32 drift at random init (tests/golden/medium_long: 0.09 on outputs of range ~5) -- on both forms of the forward; what it
    guards is a wrong PATH at an odd shape, not rounding."""
    case = CASES[ci]
    shp, sd, batch, cfg = _mk(orc, tok, case, 100 + ci)
    x = batch[:, :-1].contiguous()
    with torch.no_grad():
        ref = orc.midi_forward(sd, shp, x)
    model = mm.MIDIModel(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda", torch.bfloat16)
    with torch.no_grad():
        h = model.forward(x.cuda()).float().cpu()
    hg = model.forward(x.cuda()).detach().float().cpu()
    bound = 0.04 * ref.abs().max().item()
    err, errg = (h - ref).abs().max().item(), (hg - ref).abs().max().item()
    assert err <= bound and errg <= bound, (case, err, errg, bound)
    # the bf16 training step on the same batch: the loss near the oracle's, a few gradients pointing the oracle's way
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss, _ = orc.training_loss(sdg, shp, batch, tok.pad_id)
    ref_loss.backward()
    tm = TrainMIDIModel(cfg, lr=1e-3, warmup=2, max_step=10, accumulate_grad_batches=1)
    tm.load_state_dict(sd, strict=True)
    tm = tm.to("cuda", torch.bfloat16)
    loss = tm.training_step(batch)
    assert abs(loss.item() - ref_loss.item()) < 0.05, (loss.item(), ref_loss.item())
    named = dict(tm.named_parameters())
    for n in ("net.layers.0.self_attn.q_proj.weight", "net.layers.1.mlp.gate_proj.weight", "net_token.layers.0.mlp.down_proj.weight",
              "lm_head.weight", "net.layers.2.input_layernorm.weight", "net.norm.weight"):
        a, b = named[n].grad.float().cpu().flatten(), sdg[n].grad.flatten()
        cos = torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)
        assert cos.item() > 0.97, (case, n, cos.item())


GEN_CASES = [  # (CASES index for the widths, batch, prompt kind, prompt events, max_len, mask options)
    (0, 1, None, 0, 20, {}), (2, 3, "shared2d", 5, 18, {}), (4, 2, "per_row", 9, 16, {}), (6, 5, "short_octets", 3, 12, {}),
    (7, 2, "one_row_3d", 4, 14, {"disable_patch_change": True, "disable_control_change": True, "disable_channels": [0, 3, 9]}),
    (3, 4, "per_row", 1, 10, {"ban_eos": True}),
]


@pytest.mark.parametrize("gi", range(len(GEN_CASES)))
def test_fp32_greedy_generate_ids_match_oracle_on_fuzzed_prompts(orc, tok, gi):
    """generate() over the prompt forms midi_model.py:171-188 accepts -- None, (n, 8) shared, (1, n, 8), (B, n, 8), octets shorter
    than 8 -- at odd batch sizes, with and without the serving loop's mask options: fp32 greedy (top_k = 1: the draw is the arg-max
    whatever the generator) ids equal the oracle's, id for id."""
    ci, B, kind, n, max_len, opts = GEN_CASES[gi]
    case = CASES[ci]
    shp, sd, _, cfg = _mk(orc, tok, case, 300 + gi)
    ev = orc.synthetic_events(tok, B, max(n, 1) + 1, seed=400 + gi).numpy()  # row 0 of each sequence is BOS
    prompt = {None: None, "shared2d": ev[0, :n], "per_row": ev[:, :n], "one_row_3d": ev[:1, :n], "short_octets": ev[:1, :n, :6]}[kind]
    ref = orc.generate(sd, shp, tok, prompt=prompt, batch_size=B, max_len=max_len, top_k=1, **opts)
    model = mm.MIDIModel(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda", torch.float32)
    out = model.generate(prompt, batch_size=B, max_len=max_len, top_k=1, **opts)
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert (out == ref).all(), (GEN_CASES[gi], np.argwhere(out != ref)[:5])
