"""Parity on PEAKED distributions (the reference's real use: trained weights, app.py:299-320).

Every other model-level GPU test runs N(0, 0.02) random-init weights: logits nearly flat, most top-2 margins inside the bf16
drift, the top-p / top-k filter never binding.  Here the model is the tiny one TRAINED with the real reference on a structured
corpus (tests/gen_golden_trained.py -> tests/golden/tiny_trained.npz: eval loss 0.56, half of the rows with p_max > 0.9, a quarter
below 0.5, median top-2 margin 2.3 against a reference bf16 logits drift of 0.08; weights exact in bf16), whose shape (hidden 256,
inner 1024 / 256) sends the bf16 decode step through the PRODUCTION kernels: captured graphs, RMSNorm folded into mh_gemm_skinny,
the fused sampler.  midi_model.py:152-165 (sample_top_p_k), :195-248 (generate).
"""
import os

import numpy as np
import pytest
import torch

import midi_model_amd as mm
from conftest import trained_config

pytestmark = pytest.mark.gpu

DRIFT = 1.5


@pytest.fixture(scope="module")
def tok():
    return mm.MIDITokenizerV2()


def ref_sampler(probs, p, k, noise):
    """midi_model.py:152-165 with the draw spelled out (multinomial = argmax(p / q), variate j on sorted rank j; oracle
    sample_top_p_k) and a STABLE sort: among exactly equal probabilities (bf16 logits produce them) the lower id ranks first,
    which is the device sampler's documented tie rule -- the reference's unstable torch.sort leaves that order unspecified."""
    ps, idx = torch.sort(probs, dim=-1, descending=True, stable=True)
    csum = torch.cumsum(ps, dim=-1)
    ps = ps.masked_fill(csum - ps > p, 0.0)
    ps[..., k:] = 0
    ps = ps / ps.sum(dim=-1, keepdim=True)
    return torch.gather(idx, -1, (ps / noise).argmax(-1, keepdim=True))[..., 0]


def build(sd, dtype):
    m = mm.MIDIModel(trained_config())
    m.load_state_dict(sd, strict=True)
    return m.to("cuda", dtype).eval()


def test_trained_fp32_greedy_generate_is_the_reference_id_for_id(trained, tok):
    """fp32 generate(), greedy, on the trained weights: THE REFERENCE'S ids (golden, produced by /root/reference itself), from
    BOS and continuing a corpus prompt -- 39 + 23 events x 4 / 2 rows, every id equal.  (Greedy needs no random stream, so the
    device and the reference's CPU generator cannot differ.)"""
    shp, sd, g = trained
    m = build(sd, torch.float32)
    out = m.generate(None, batch_size=4, max_len=40, top_k=1)
    assert out.shape == g["greedy_b4"].shape and (out == g["greedy_b4"]).all(), np.argwhere(out != g["greedy_b4"])[:4]
    out = m.generate(g["prompt"], batch_size=2, max_len=36, top_k=1)
    assert (out == g["prompt_greedy_b2"]).all(), np.argwhere(out != g["prompt_greedy_b2"])[:4]


def test_trained_forward_and_loss(orc, trained, tok):
    """forward / forward_token on the trained weights against the reference's outputs on held-out corpus rows: fp32 rtol 1e-3 on
    the logits statistics and the arg-max wherever the reference's margin exceeds 1e-3; bf16 inside 1.5x the reference's own
    bf16 drift (logits, hidden), arg-max equal wherever the margin exceeds twice that bound -- and on these weights that is most
    positions (>= 0.8 required; random-init fixtures reach 0.4)."""
    from midi_model_amd.train import TrainMIDIModel
    shp, sd, g = trained
    ev = torch.from_numpy(g["eval_batch"])
    x, y = ev[:, :-1].contiguous(), ev[:, 1:].contiguous()
    with torch.no_grad():
        hid_o = orc.midi_forward(sd, shp, x)
        log_o = orc.midi_forward_token(sd, shp, hid_o.reshape(-1, hid_o.shape[-1]), y.reshape(-1, 8)[:, :-1])
    margin = torch.from_numpy(g["eval_logits_margin"])
    amax = torch.from_numpy(g["eval_logits_argmax"])
    for dtype in (torch.float32, torch.bfloat16):
        m = build(sd, dtype)
        with torch.no_grad():
            hid = m.forward(x.cuda())
            logits = m.forward_token(hid.reshape(-1, hid.shape[-1]), y.reshape(-1, 8)[:, :-1].cuda()).float().cpu()
        hid = hid.float().cpu()
        if dtype == torch.float32:
            assert ((hid - hid_o).abs() <= 1e-3 * hid_o.abs() + 2e-4).all()
            assert ((logits - log_o).abs() <= 1e-3 * log_o.abs() + 5e-4).all()
            safe = margin > 1e-3
        else:
            hb, lb = DRIFT * float(g["ref_bf16_hidden_maxerr"]), DRIFT * float(g["ref_bf16_logits_maxerr"])
            eh, el = (hid - hid_o).abs().max().item(), (logits - log_o).abs().max().item()
            assert eh <= hb and el <= lb, (eh, hb, el, lb)
            safe = margin > 2 * lb
            assert safe.float().mean() > 0.8, safe.float().mean()
        assert (logits.argmax(-1)[safe] == amax[safe]).all()
        t = TrainMIDIModel(trained_config())
        t.load_state_dict(sd)
        t = t.to("cuda", dtype)
        loss = t.training_step(ev).item()
        want = float(g["eval_loss"])
        assert abs(loss - want) < (2e-4 if dtype == torch.float32 else 3e-2), (loss, want, float(g["ref_bf16_loss"]))


@pytest.mark.parametrize("dtype,top_k", [(torch.float32, 20), (torch.bfloat16, 20), (torch.bfloat16, 1)], ids=["fp32_sampled", "bf16_sampled", "bf16_greedy"])
def test_trained_decode_session_all_64_rows(orc, trained, tok, dtype, top_k):
    """The production decode session (batch 64, graphs; bf16: folded norms + mh_gemm_skinny + fused sampler) on the trained
    weights, a 13-event corpus prompt then 8 decoded events, ALL 64 rows followed by the oracle (teacher-forced with the
    device's ids), sampled with top_p 0.98 / top_k 20 at temperature 1 and greedy:
      * logits within the dtype's bound of the oracle's cached forward at every token step;
      * the id the device drew IS the reference's sampler (sort, cumulative top-p cut, top-k cut, renormalise, argmax(p / q):
        midi_model.py:152-165) applied to the device's own logits with the device's own Exp(1) variates -- every row, every
        step: on these weights the top-p cut removes most of the vocabulary and often leaves 1-3 ids;
      * fp32: it is also what that sampler draws from the ORACLE's logits on the same variates (>= 99 % of the draws; an fp32
        logit difference of 1e-5 can move a draw only at a near-tie of p / q);
      * greedy bf16: the id equals the oracle's grammar-masked arg-max wherever its top-2 margin exceeds twice the bound, and that
        is >= 85 % of the sampling positions here (measured 87 %; the random-init session test can only require 40 %).  The rest
        are the corpus's DESIGNED open choices -- four equiprobable velocities, twelve motifs to start -- where the reference's
        own top-2 margin is ~0: 88 % of the golden's held-out positions clear the bound, and longer training does not move that
        (margins and the reference's bf16 drift grow together; tried: loss 0.25, median margin 7.0, drift 0.20, still 88 %)."""
    from midi_model_amd.decode import DecodeSession
    shp, sd, g = trained
    model = build(sd, dtype)
    B, n_events, cap, V = 64, 8, 256, tok.vocab_size
    temp, top_p = 1.0, 0.98
    log_bound = DRIFT * float(g["ref_bf16_logits_maxerr"]) if dtype == torch.bfloat16 else 2e-2   # (fp32: rtol 1e-3 of logits up to ~20)
    prompt = torch.from_numpy(g["prompt"])[None].repeat(B, 1, 1)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    with torch.inference_mode():
        ses = DecodeSession(model, B, cap, temp, top_p, top_k)
        if dtype == torch.bfloat16:
            assert ses.g_net is not None and ses.g_steps is not None, "captured graphs are the production form"
            assert ses.fold1 is not None and ses.lm_fold is not None and ses.fused_sampler, "folded norms + fused sampler"
        ses.first_mask.copy_(model._grammar()[0])
        ses.ban.zero_()
        ses.reset()
        ses.begin(torch.Generator(device="cuda").manual_seed(11))
        ses.prefill(prompt.cuda())
        cache1 = orc.KV()
        hid_o = orc.midi_forward(sd, shp, prompt, cache1)[:, -1]
        worst, draws, same_as_oracle, checked, total, narrow = 0.0, 0, 0, 0, 0, 0
        for ev_i in range(n_events):
            cache2 = orc.KV()
            names, end = [""] * B, [False] * B
            n_steps, i, prev = tok.max_token_seq, 0, None
            while i < n_steps:
                ses.tok_step(i)
                lg = ses.logits[:, :V].float().cpu()
                ids = ses.seq[:, i].cpu()
                q = ses.q_all[i].cpu()
                lo = orc.midi_forward_token(sd, shp, hid_o if i == 0 else None, None if i == 0 else prev[:, None], cache2)[:, -1]
                e = (lg - lo).abs().max().item()
                worst = max(worst, e)
                assert e <= log_bound, (ev_i, i, e, log_bound)
                mask = orc.grammar_mask(tok, i, names, end)
                assert mask.bool().gather(1, ids[:, None]).all(), "the device sampled an id outside the grammar mask"
                if ses.fused_sampler:
                    # the reference's chain on the DEVICE's logits and variates (variate j belongs to sorted rank j)
                    pd = torch.softmax(lg / temp, -1) * mask
                    want = ref_sampler(pd, top_p, top_k, q)
                    assert (ids == want).all(), (ev_i, i, (ids != want).nonzero().flatten().tolist())
                    po = torch.softmax(lo / temp, -1) * mask
                    want_o = ref_sampler(po, top_p, top_k, q)
                    draws += B
                    same_as_oracle += int((ids == want_o).sum())
                    ps = torch.sort(po / po.sum(-1, keepdim=True), -1, descending=True).values
                    narrow += int((((ps.cumsum(-1) - ps) <= top_p).sum(-1) <= 3).sum())  # rows whose nucleus is <= 3 ids
                legal = lo.masked_fill(~mask.bool(), float("-inf"))
                top2 = legal.topk(2, -1)
                margin = top2.values[:, 0] - top2.values[:, 1]
                if top_k == 1:
                    safe = margin > 2 * log_bound
                    total += B
                    checked += int(safe.sum())
                    assert (ids[safe] == top2.indices[:, 0][safe]).all(), (ev_i, i)
                if i == 0:
                    names = [tok.id_events.get(int(t), "") for t in ids]
                    end = [int(t) == tok.eos_id for t in ids]
                    alive = [len(tok.events[n]) for n, e_ in zip(names, end) if not e_]
                    n_steps = 2 if not alive else (alive[0] + 1 if all(a == alive[0] for a in alive) else tok.max_token_seq)
                prev = ids
                i += 1
            ses.consumed(n_steps)
            event = ses.seq.cpu().clone()
            ses.net_step()
            hid_o = orc.midi_forward(sd, shp, event[:, None, :], cache1)[:, -1]
        ses.end()
    print(f"trained decode session [{dtype}, top_k {top_k}]: worst logits err {worst:.4f} (bound {log_bound:.4f}); "
          f"draws equal to the oracle's {same_as_oracle}/{draws}, rows with a nucleus of <= 3 ids {narrow}/{draws}; greedy checked {checked}/{total}")
    if top_k == 1:
        assert checked >= 0.85 * total, (checked, total)
    else:
        assert narrow > 0.3 * draws, "the top-p filter is supposed to bind on these weights"
        if dtype == torch.float32:
            assert same_as_oracle >= 0.99 * draws, (same_as_oracle, draws)
