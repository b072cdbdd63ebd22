"""The PRODUCTION decode path against the oracle on the MI355X: bf16 tv2o-medium through ``DecodeSession`` exactly as
``generate()`` drives it -- captured hipGraphs, RMSNorm weights folded into the projections, the decode-step kernels, the
fused sampler inside the graphs -- compared with the oracle's KV-cached forward (midi_model.py:195-248) on the same tokens.

Also here: the serving-loop mask options (app.py:73-86) against the oracle, the reference's OWN serving loop
(tests/ref_loops.py, verbatim app.py:27-120) driving our module with a real ``transformers.DynamicCache``, generation after a
training step in the same process (pooled sessions must follow the weights), checkpoint round trips on the device.
"""
import os

import numpy as np
import pytest
import torch

import midi_model_amd as mm
from midi_model_amd.train import TrainMIDIModel

pytestmark = pytest.mark.gpu

DRIFT = 1.5


@pytest.fixture(scope="module")
def tok():
    return mm.MIDITokenizerV2()


@pytest.fixture(scope="module")
def medium_bf16(orc, tok):
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    m = mm.MIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium"))
    m.load_state_dict(sd, strict=True)
    return shp, sd, m.to("cuda", torch.bfloat16).eval()


def _n_steps(tok, ids):
    """the reference's break rule (midi_model.py:232-235) from the event ids sampled at position 0"""
    alive = [len(tok.events[tok.id_events[t]]) for t in ids if t != tok.eos_id]
    if not alive:
        return 2
    return alive[0] + 1 if all(a == alive[0] for a in alive) else tok.max_token_seq


@pytest.mark.parametrize("B,P,n_events,cap,rows", [(4, 65, 8, 256, None), (64, 17, 4, 256, None),
                                                      (64, 1001, 8, 2048, (0, 9, 18, 27, 36, 45, 54, 63)),
                                                      (64, 1001, 1, 2048, None)],
                         ids=["b4", "b64_benchmarked_batch", "b64_cap2048_depth1000_benchmarked_session",
                              "b64_cap2048_depth1000_all_64_rows_one_event"])
def test_production_decode_session_matches_oracle(orc, tok, medium_bf16, golden, B, P, n_events, cap, rows):
    """(B = 64: the batch bench.py --mode generate runs, where mh_gemm_skinny takes its 64-row tilings.  The third case is the
    session bench.py --mode generate checks out -- BASELINE configs[3]: capacity 2048, batch 64 -- at the DEPTH it runs to:
    a 1001-event prefill, then 8 events decoded by the replayed graphs with pos_dev = 1001..1008, i.e. attn_decode_kernel<64>
    over 1000+ cached keys at 64 x 16 heads.  Sequences are independent, so the oracle follows 8 of the 64 rows (every row
    of the device batch still goes through the 64-row tilings; midi_model.py:195-248).  The fourth case follows ALL 64 rows
    through one event at that depth -- one oracle prefill of 64 x 1001 events.)
    64-event prompt prefill + 8 decoded events, greedy (top_k = 1): after every replayed graph the session's hidden
    state / logits are within the reference's own bf16 drift (x1.5) of the oracle's cached fp32 forward on the SAME
    tokens, and the greedy id equals the oracle's masked arg-max wherever the oracle's top-2 margin exceeds twice that
    bound.  The oracle is teacher-forced with the ids the device picked, so both sides see identical inputs throughout."""
    from midi_model_amd.decode import DecodeSession
    shp, sd, model = medium_bf16
    g = golden("medium_long_S2048.npz")
    hid_bound = DRIFT * float(g["ref_bf16_hidden_maxerr"])
    log_bound = DRIFT * float(g["ref_bf16_logits_maxerr"])
    V = tok.vocab_size
    prompt = orc.synthetic_events(tok, B, P, seed=31)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    with torch.inference_mode():
        ses = DecodeSession(model, B, cap, 1.0, 0.98, 1)
        assert ses.cap == cap and ses.kv1.k.shape[-2] == cap
        assert ses.g_net is not None and ses.g_steps is not None and ses.g_noise is not None, "captured graphs are the production form"
        assert ses.fold1 is not None and ses.lm_fold is not None and ses.fused_sampler, "folded norms + fused sampler"
        ses.first_mask.copy_(model._grammar()[0])
        ses.ban.zero_()
        ses.reset()
        ses.begin(torch.Generator(device="cuda").manual_seed(1))
        ses.prefill(prompt.cuda())
        R = torch.arange(B) if rows is None else torch.tensor(rows)  # the rows the oracle follows
        Bo = R.numel()
        cache1 = orc.KV()
        hid_o = orc.midi_forward(sd, shp, prompt[R], cache1)[:, -1]
        worst_h, worst_l, checked, total = 0.0, 0.0, 0, 0
        for ev_i in range(n_events):
            assert int(ses.pos.item()) == P + ev_i
            err = (ses.hidden.float().cpu()[R] - hid_o).abs().max().item()
            worst_h = max(worst_h, err)
            assert err <= hid_bound, (ev_i, err, hid_bound)
            cache2 = orc.KV()
            names, end = [""] * Bo, [False] * Bo
            n_steps, i, prev = tok.max_token_seq, 0, None
            while i < n_steps:
                ses.tok_step(i)
                lg = ses.logits[:, :V].float().cpu()[R]
                ids_all = ses.seq[:, i].cpu()
                ids = ids_all[R]
                lo = orc.midi_forward_token(sd, shp, hid_o if i == 0 else None, None if i == 0 else prev[:, None], cache2)[:, -1]
                e = (lg - lo).abs().max().item()
                worst_l = max(worst_l, e)
                assert e <= log_bound, (ev_i, i, e, log_bound)
                mask = orc.grammar_mask(tok, i, names, end).bool()
                legal = lo.masked_fill(~mask, float("-inf"))
                top2 = legal.topk(2, -1)
                margin = top2.values[:, 0] - top2.values[:, 1]  # (inf where a single id is legal)
                assert mask.gather(1, ids[:, None]).all(), "the device sampled an id outside the grammar mask"
                safe = margin > 2 * log_bound
                total += Bo
                checked += int(safe.sum())
                assert (ids[safe] == top2.indices[:, 0][safe]).all(), (ev_i, i, ids.tolist(), top2.indices[:, 0].tolist())
                if i == 0:
                    assert torch.equal(ses.ev.cpu(), ids_all)
                    names = [tok.id_events.get(int(t), "") for t in ids]
                    end = [int(t) == tok.eos_id for t in ids]
                    n_steps = _n_steps(tok, ids_all.tolist())  # (the break rule looks at the whole device batch)
                prev = ids
                i += 1
            ses.consumed(n_steps)  # the draws the reference loop would have made for this event (decode.py contract)
            event = ses.seq.cpu().clone()[R]
            for b in range(Bo):  # positions past the event's arity hold PAD (fill_rest of the fused sampler)
                ar = 0 if end[b] else len(tok.events[names[b]])
                assert (event[b, 1 + ar:] == tok.pad_id).all()
            ses.net_step()
            hid_o = orc.midi_forward(sd, shp, event[:, None, :], cache1)[:, -1]
        ses.end()
    print(f"decode session vs oracle: worst hidden err {worst_h:.4f} (bound {hid_bound:.4f}), worst logits err {worst_l:.4f} "
          f"(bound {log_bound:.4f}); greedy ids checked on {checked}/{total} rows with a safe margin")
    assert checked > 0.4 * total


def test_greedy_generate_600_events_fp32_follows_the_oracle_id_for_id(orc, tok):
    """``generate()`` itself at depth (midi_model.py:195-248): fp32 tv2o-medium, greedy (top_k = 1), BOS prompt, batch 2,
    max_len 600 -- a capacity-1024 session, ~600 replays of the net / steps graphs with pos_dev running past 255 and 511.
    The oracle is teacher-forced with the device's ids in ONE uncached pass (event-level forward over the 599 generated
    events, token-level forward over every octet): at every sampling position the device's id must be the oracle's
    grammar-masked arg-max wherever the oracle's top-2 margin exceeds 1e-3 (fp32 noise is ~1e-5: nearly every position
    qualifies, and a position that does not cannot hide a wrong continuation because the oracle sees the device's tokens)."""
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    m = mm.MIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium"))
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda", torch.float32).eval()
    B, L, T, V = 2, 600, tok.max_token_seq, tok.vocab_size
    out = m.generate(None, batch_size=B, max_len=L, top_k=1, ban_eos=True, generator=torch.Generator(device="cuda").manual_seed(3))
    assert out.shape == (B, L, T) and out.dtype == np.int64
    ses = m._sessions.idle[-1]
    assert ses.cap == 1024 and int(ses.pos.item()) == L - 1, (ses.cap, int(ses.pos.item()))
    ids = torch.from_numpy(out)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    with torch.inference_mode():
        hidden = orc.midi_forward(sd, shp, ids[:, :-1])                        # (B, L-1, D): hidden i predicts event i+1
        tgt = ids[:, 1:].reshape(B * (L - 1), T)
        logits = orc.midi_forward_token(sd, shp, hidden.reshape(B * (L - 1), -1), tgt[:, :-1])  # (N, 8, V)
    N = tgt.shape[0]
    names = [tok.id_events[int(t)] for t in tgt[:, 0]]
    bad, checked = [], 0
    for i in range(T):
        mask = orc.grammar_mask(tok, i, names if i else [""] * N, [False] * N, ban_eos=True).bool()
        assert mask.gather(1, tgt[:, i:i + 1]).all(), f"position {i}: an id outside the grammar mask"
        legal = logits[:, i].masked_fill(~mask, float("-inf"))
        top2 = legal.topk(2, -1)
        margin = top2.values[:, 0] - top2.values[:, 1]
        safe = margin > 1e-3
        checked += int(safe.sum())
        wrong = (tgt[:, i] != top2.indices[:, 0]) & safe
        bad += [(int(n), i) for n in wrong.nonzero().flatten()]
    assert not bad, f"device ids differ from the oracle's arg-max at (row*event, position): {bad[:8]}"
    assert checked > 0.97 * N * T, (checked, N * T)
    print(f"greedy generate, {L} events x {B}: {checked}/{N * T} sampling positions checked against the oracle's arg-max, all equal")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_chunked_continuation_of_a_cached_forward(orc, tok, golden, dtype):
    """``MIDIModel.forward(x, cache)`` with a NON-EMPTY cache and q_len > 1 (midi_model.py:137-150; chunked prefill): tv2o-medium,
    a 300-event prefill, a 212-event chunk (its first query tile starts below the cached length: 300 is not a multiple of 128), a
    129-event chunk, then one event -- against the oracle's cached forward fed the same chunks, and against the uncached forward
    over all 642 events.  fp32: rtol 1e-3 on every hidden state; bf16: within 1.5x the reference's own bf16 drift."""
    from midi_model_amd.engine import KVState
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    m = mm.MIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium"))
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda", dtype).eval()
    B, cuts = 2, (0, 300, 512, 641, 642)
    x = orc.synthetic_events(tok, B, cuts[-1], seed=91)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))

    class AnyCache:
        pass

    cache, kv_o = AnyCache(), orc.KV()
    outs, outs_o = [], []
    with torch.no_grad():
        for a, b in zip(cuts, cuts[1:]):
            outs.append(m.forward(x[:, a:b].cuda(), cache=cache).float().cpu())
            outs_o.append(orc.midi_forward(sd, shp, x[:, a:b], kv_o))
        full_o = orc.midi_forward(sd, shp, x)
    assert isinstance(cache._mh_state, KVState) and cache._mh_state.len == cuts[-1]
    got, want = torch.cat(outs, 1), torch.cat(outs_o, 1)
    assert (want - full_o).abs().max() < 2e-4          # the oracle's cached chunks = its uncached forward
    err = (got - want).abs()
    if dtype == torch.float32:
        assert (err <= 1e-3 * want.abs() + 3e-4).all(), err.max().item()
    else:
        g = golden("medium_long_S2048.npz")
        bound = DRIFT * float(g["ref_bf16_hidden_maxerr"])
        assert err.max().item() <= bound, (err.max().item(), bound)
        assert err.pow(2).mean().sqrt().item() <= DRIFT * float(g["ref_bf16_hidden_rmserr"])


def test_fused_sampler_inside_graphs_equals_oracle_chain_on_same_noise(orc, tok, medium_bf16):
    """Seeded sampling through the captured graphs (temp 1, top_p 0.98, top_k 20): after every token step, feed the
    oracle's sampling chain (softmax * mask -> sort -> top-p -> top-k -> renormalise -> argmax(p / q), midi_model.py:152-165
    + 216-218) the logits the step produced and the Exp(1) variates the step drew (the session's noise buffer): the ids must be
    the ones the fused device sampler wrote."""
    from midi_model_amd.decode import DecodeSession
    shp, sd, model = medium_bf16
    B, V = 8, tok.vocab_size
    prompt = orc.synthetic_events(tok, B, 9, seed=33)
    with torch.inference_mode():
        ses = DecodeSession(model, B, 256, 1.0, 0.98, 20)
        assert ses.fused_sampler and ses.g_steps is not None and ses.g_noise is not None
        ses.first_mask.copy_(model._grammar()[0])
        ses.ban.zero_()
        ses.reset()
        ses.begin(torch.Generator(device="cuda").manual_seed(7))
        ses.prefill(prompt.cuda())
        n = 0
        for ev_i in range(6):
            names, end = [""] * B, [False] * B
            n_steps, i = tok.max_token_seq, 0
            ses.draw_noise()  # the event's Exp(1) variates (the noise graph, on its side stream)
            while i < n_steps:
                ses.tok_step(i)
                lg, q, ids = ses.logits[:, :V].float().cpu(), ses.q_all[i].cpu(), ses.seq[:, i].cpu()
                mask = orc.grammar_mask(tok, i, names, end)
                scores = torch.softmax(lg / 1.0, dim=-1) * mask
                want = orc.sample_top_p_k(scores[:, None], 0.98, 20, noise=q[:, None])[:, 0]
                diff = (want != ids).nonzero().flatten().tolist()
                for b in diff:  # only an exact probability tie may order two ids differently
                    assert scores[b, want[b]] == scores[b, ids[b]], (ev_i, i, b, int(want[b]), int(ids[b]))
                n += B - len(diff)
                if i == 0:
                    names = [tok.id_events.get(int(t), "") for t in ids]
                    end = [int(t) == tok.eos_id for t in ids]
                    n_steps = _n_steps(tok, ids.tolist())
                    assert ses.n_steps_of(ids.tolist())[0] == n_steps
                i += 1
            ses.consumed(n_steps)
            ses.net_step()
        ses.end()
    assert n > 100


def test_steps_graph_equals_step_by_step_and_rng_stream_is_the_reference_loops(orc, tok, medium_bf16):
    """generate()'s form -- ONE graph for the 8 token steps on noise drawn ahead by the noise graph, generator wound back to
    the draws the reference loop makes -- against the step-by-step form on the same noise (same kernels: identical tokens),
    and the generator state it hands back against the count of sampling calls the reference's break rule gives."""
    from midi_model_amd.decode import DecodeSession
    shp, sd, model = medium_bf16
    B = 6
    prompt = orc.synthetic_events(tok, B, 5, seed=35).cuda()
    gen = torch.Generator(device="cuda")
    with torch.inference_mode():
        ses = DecodeSession(model, B, 256, 1.0, 0.98, 20)
        ses.first_mask.copy_(model._grammar()[0])
        ses.ban.zero_()
        events, calls = [], 0
        ses.reset()
        ses.begin(gen.manual_seed(11))
        off0 = gen.get_offset()
        ses.prefill(prompt)
        for _ in range(5):
            ev, end_all = ses.sample_event()
            calls += ses.n_steps_of(ev[:, 0].tolist())[0]
            events.append(ev)
            ses.draw_noise()
            ses.net_step()
        ses.end()
        assert gen.get_offset() == off0 + calls * ses._draw_inc, "the generator stands exactly `calls` draws on"
        # the same five events, token step by token step, on the same seed
        ses.reset()
        ses.begin(gen.manual_seed(11))
        ses.prefill(prompt)
        for k in range(5):
            ses.draw_noise()
            for i in range(tok.max_token_seq):
                ses.tok_step(i)
            assert (ses.seq.cpu().numpy() == events[k]).all(), k
            ses.consumed(ses.n_steps_of(events[k][:, 0].tolist())[0])
            ses.net_step()
        ses.end()
        # the contract is enforced: an event stepped with tok_step and never reported with consumed() cannot be followed by
        # another event's step 0 on the same (already read) variates
        ses.begin(gen.manual_seed(11))
        ses.tok_step(0)
        with pytest.raises(RuntimeError, match="consumed"):
            ses.tok_step(0)
        ses.consumed(1)
        ses.tok_step(0)
        ses.consumed(1)
        ses.end()


def test_generate_with_mask_options_equals_oracle(orc, tok):
    """fp32, greedy: generate() / generate_stream() with disable_patch_change / disable_control_change / disable_channels and
    the 4096-event prompt crop reproduce the oracle's ids (oracle.generate restates app.py:27-120 with the same options)."""
    cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512)
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    model = mm.MIDIModel(cfg)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    kw = dict(disable_patch_change=True, disable_control_change=True, disable_channels=[0, 3, 9])
    for seed, B in ((3, 3), (4, 2)):
        prompt = orc.synthetic_events(tok, 1, 6, seed=seed)[0].numpy()
        want = orc.generate(sd, shp, tok, prompt, batch_size=B, max_len=22, top_k=1, **kw)
        got = model.generate(prompt, batch_size=B, max_len=22, top_k=1, **kw)
        assert got.shape == want.shape and (got == want).all()
        evs = np.stack(list(model.generate_stream(prompt, batch_size=B, max_len=22, top_k=1, **kw)), 1)
        assert (evs == want[:, 6:]).all()
    banned = [tok.parameter_ids["channel"][c] for c in (0, 3, 9)]
    assert not np.isin(want, banned).any()
    # no options: a different stream (the masks matter for these weights)
    plain = orc.generate(sd, shp, tok, prompt, batch_size=2, max_len=22, top_k=1)
    assert (model.generate(prompt, batch_size=2, max_len=22, top_k=1) == plain).all()
    # prompt crop (app.py:53): a 4100-event prompt is cut to its last 4096 events before anything else happens
    long_prompt = orc.synthetic_events(tok, 1, 4100, seed=9)[0].numpy()
    want = orc.generate(sd, shp, tok, long_prompt, batch_size=1, max_len=4099, top_k=1, crop=4096)
    evs = np.stack(list(model.generate_stream(long_prompt, batch_size=1, max_len=4099, top_k=1)), 1)
    assert evs.shape == (1, 3, 8) and (evs == want[:, 4096:]).all()


def test_reference_serving_loop_runs_on_the_drop_in(orc, tok):
    """app.py:54-120 verbatim (tests/ref_loops.py) with a REAL transformers.DynamicCache the caller creates, and the
    trainer-mixin MRO of train.py:106, against our module: greedy ids equal the oracle's; with a seeded CUDA generator the
    loop's stream equals generate()'s (same draws from the same generator state)."""
    from transformers import DynamicCache
    import midi_model  # the drop-in module name (app.py:20, train.py:21)
    from ref_loops import serving_loop

    class Mixin:  # stands in for pl.LightningModule in ``class TrainMIDIModel(MIDIModel, pl.LightningModule)``
        def __init__(self, *a, **k):
            super().__init__()
            self.mixin_ready = True

    class T(midi_model.MIDIModel, Mixin):
        def __init__(self, config):
            super().__init__(config=config)

    assert midi_model.MIDIModel is mm.MIDIModel and T.__mro__[1] is mm.MIDIModel
    cfg = midi_model.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512)
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    model = T(cfg)
    assert len(list(model.named_parameters())) == len(sd)
    model.load_state_dict(sd, strict=False)
    model = model.to(device="cuda", dtype=torch.float32).eval()
    prompt = orc.synthetic_events(tok, 1, 5, seed=12)[0].numpy()
    got = np.stack(list(serving_loop(model, model.tokenizer, DynamicCache, prompt, batch_size=2, max_len=16, top_k=1,
                                     disable_control_change=True, disable_channels=[9])), 1)
    want = orc.generate(sd, shp, tok, prompt, batch_size=2, max_len=16, top_k=1, disable_control_change=True,
                        disable_channels=[9])
    assert (got == want[:, 5:]).all()
    gen = torch.Generator(device="cuda")
    a = np.stack(list(serving_loop(model, model.tokenizer, DynamicCache, None, batch_size=3, max_len=12,
                                   generator=gen.manual_seed(5))), 1)
    b = model.generate(None, batch_size=3, max_len=12, generator=gen.manual_seed(5))
    assert (a == b[:, 1:]).all()


def test_generate_after_training_step_follows_the_weights(orc, tok):
    """pooled decode sessions hold folded copies (norm weight x projection) of the parameters; AdamW
    writes the parameters through raw pointers.  generate -> fit_step -> generate in one process must equal a freshly
    built model with the trained weights (the reference's gen_example every validation epoch has this shape)."""
    cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 1024)  # (token-level width 256: the folded form is active)
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=1024, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=2)
    model = TrainMIDIModel(cfg, lr=3e-2, warmup=0, max_step=100, accumulate_grad_batches=1)
    model.load_state_dict(sd)
    model = model.to("cuda", torch.bfloat16)
    kw = dict(batch_size=3, max_len=24, top_k=1, ban_eos=True)
    first = model.generate(None, **kw)
    assert len(model._sessions.idle) == 1 and model._sessions.idle[0].fold1 is not None
    ses = model._sessions.idle[0]
    for s in range(3):
        model.fit_step(orc.synthetic_events(tok, 2, 33, seed=50 + s).cuda())
    second = model.generate(None, **kw)
    assert model._sessions.idle[0] is ses, "the pooled session was reused"
    fresh = mm.MIDIModel(cfg)
    fresh.load_state_dict({k: v.detach().float().cpu() for k, v in model.state_dict().items()})
    fresh = fresh.to("cuda", torch.bfloat16)
    want = fresh.generate(None, **kw)
    assert (second == want).all()
    assert (first != second).any(), "three steps at lr 3e-2 were meant to change the greedy stream"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_checkpoint_round_trips_on_device(orc, tok, tmp_path, dtype):
    """SURVEY f3 (app.py:311-316, train.py:246-270,433-436): save_pretrained -> from_pretrained, a Lightning-shaped
    ``{"state_dict": ...}`` .ckpt and a bare safetensors file all land in the flat buffer / fused q|k|v and gate|up views and
    give bit-identical forward outputs; a checkpoint that misses tensors raises instead of leaving random weights."""
    from safetensors.torch import save_file
    cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 512)
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=5)
    model = mm.MIDIModel(cfg)
    model.load_state_dict(sd)
    model = model.to("cuda", dtype).eval()
    x = orc.synthetic_events(tok, 2, 12, seed=3).cuda()
    with torch.no_grad():
        want = model.forward(x).clone()
    model.save_pretrained(str(tmp_path / "hf"))
    assert sorted(os.listdir(tmp_path / "hf")) == ["config.json", "model.safetensors"]
    state = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    assert len(state) == 50 and state["net.layers.0.self_attn.q_proj.weight"].dtype == dtype
    torch.save({"state_dict": state, "epoch": 3, "global_step": 17}, str(tmp_path / "last.ckpt"))
    save_file(state, str(tmp_path / "model.safetensors"))
    torch.save({"state_dict": {"model." + k: v for k, v in state.items()}}, str(tmp_path / "prefixed.ckpt"))
    loaded = [mm.MIDIModel.from_pretrained(str(tmp_path / "hf")),
              mm.MIDIModel.from_checkpoint(cfg, str(tmp_path / "last.ckpt")),
              mm.MIDIModel.from_checkpoint(cfg, str(tmp_path / "model.safetensors")),
              mm.MIDIModel.from_checkpoint(cfg, str(tmp_path / "prefixed.ckpt"))]
    for m in loaded:
        m = m.to("cuda", dtype).eval()
        assert m._flat.dtype == dtype and m._flat.is_cuda
        lw = m._W["net"].layers[1]
        q = dict(m.named_parameters())["net.layers.1.self_attn.q_proj.weight"]
        assert lw.wqkv.data_ptr() == q.data_ptr() and lw.wqkv.shape == (768, 256), "q|k|v is one fused view of the flat buffer"
        assert torch.equal(lw.wqkv[512:].cpu(), state["net.layers.1.self_attn.v_proj.weight"])
        assert torch.equal(lw.wgu[512:].cpu(), state["net.layers.1.mlp.up_proj.weight"])
        with torch.no_grad():
            assert torch.equal(m.forward(x), want)
    # the reference's own call shape (app.py:313-316): load_state_dict(state_dict, strict=False) after torch.load
    m = mm.MIDIModel(cfg)
    res = m.load_state_dict(torch.load(str(tmp_path / "last.ckpt"), map_location="cpu", weights_only=True)["state_dict"], strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    partial = dict(state)
    del partial["net.norm.weight"]
    save_file(partial, str(tmp_path / "partial.safetensors"))
    with pytest.raises(RuntimeError, match="missing"):
        mm.MIDIModel.from_checkpoint(cfg, str(tmp_path / "partial.safetensors"))
