"""torch.autograd nodes that make ``MIDIModel.forward`` / ``forward_token`` differentiable for callers that
drive the reference's own step (``loss = F.cross_entropy(model.forward_token(model.forward(x), ...))`` then
``loss.backward()``, train.py:168-188).  Each node wraps the explicit forward/backward schedules of
``engine.py``; the fused training step in ``train.py`` calls those schedules directly instead and never
materialises the logits.
"""
from __future__ import annotations

import torch

from . import engine, ops


def _grads_for(model, pre: str, tmp_flat: torch.Tensor):
    """per-parameter gradient views (in `_stack_params` order) of a temporary flat buffer"""
    out = []
    for p, name in zip(model._stack_params(pre), _names(model, pre)):
        off, n, _ = model._offsets[name]
        out.append(tmp_flat[off:off + n].view(p.shape))
    return out


def _names(model, pre: str):
    st = getattr(model, pre)
    names = [f"{pre}.embed_tokens.weight"]
    for i in range(len(st.layers)):
        b = f"{pre}.layers.{i}."
        names += [b + "self_attn.q_proj.weight", b + "self_attn.k_proj.weight", b + "self_attn.v_proj.weight",
                  b + "self_attn.o_proj.weight", b + "mlp.gate_proj.weight", b + "mlp.up_proj.weight",
                  b + "mlp.down_proj.weight", b + "input_layernorm.weight", b + "post_attention_layernorm.weight"]
    names.append(f"{pre}.norm.weight")
    return names


class NetFn(torch.autograd.Function):
    """hidden = net(sum of the octet's embeddings)   (midi_model.py:137-150)"""

    @staticmethod
    def forward(ctx, model, tokens, *params):
        B, S, T = tokens.shape
        spec = model._specs["net"]
        W = model._W["net"]
        e = torch.empty((B * S, spec.D), dtype=model.dtype, device=model.device)
        ops.embed_sum_fwd(tokens.view(B * S, T), W.embed, e)
        need = any(ctx.needs_input_grad)
        y, saved = engine.stack_forward(spec, W, e, B, S, model.rope("net"), save=need)
        ctx.model, ctx.saved, ctx.tokens = model, saved, tokens
        return y.view(B, S, spec.D)

    @staticmethod
    def backward(ctx, dy):
        model, tokens = ctx.model, ctx.tokens
        if ctx.saved is None:
            raise RuntimeError("forward ran without saving activations (no_grad); cannot backpropagate")
        B, S, T = tokens.shape
        spec = model._specs["net"]
        tmp = torch.empty_like(model._flat)
        G = model._stack_views("net", tmp)
        dy2 = dy.contiguous().view(B * S, spec.D).to(model.dtype)
        dx = engine.stack_backward(spec, model._W["net"], G, ctx.saved, dy2, model.rope("net"), False)
        ctx.saved = None
        acc = torch.zeros(G.embed.shape, dtype=torch.float32, device=model.device)
        ops.embed_scatter_bwd(tokens.view(B * S, T), T, dx, 1, 0, 0, acc, model.tokenizer.pad_id)
        ops.cast_from_f32(acc, G.embed, False)
        return (None, None, *_grads_for(model, "net", tmp))


class TokFn(torch.autograd.Function):
    """logits = lm_head(net_token([hidden ; embed(x)]))   (midi_model.py:116-135)"""

    @staticmethod
    def forward(ctx, model, hidden, x, *params):
        spec = model._specs["net_token"]
        W = model._W["net_token"]
        V, Vp = model.tokenizer.vocab_size, model.vocab_padded
        if hidden is None:
            raise NotImplementedError("forward_token without hidden_state needs a cache (decode path)")
        N = hidden.shape[0]
        t = 0 if x is None else x.shape[1]
        T = 1 + t
        hid = hidden.to(model.dtype).contiguous()
        seq = torch.empty((N, T, spec.D), dtype=model.dtype, device=model.device)
        if x is None:
            x = torch.zeros((N, 1), dtype=torch.long, device=model.device)
        else:
            x = x.contiguous()
        ops.concat_tok_fwd(hid, x, W.embed, seq, T)
        need = any(ctx.needs_input_grad)
        h, saved = engine.stack_forward(spec, W, seq.view(N * T, spec.D), N, T, model.rope("net_token"), save=need)
        logits = torch.empty((N * T, Vp), dtype=model.dtype, device=model.device)
        ops.gemm_nt(h, model.lm_head.weight.data, logits[:, :V])
        ctx.model, ctx.saved, ctx.x, ctx.h, ctx.dims = model, saved, x, (h if need else None), (N, T, t)
        return logits.view(N, T, Vp)[:, :, :V]

    @staticmethod
    def backward(ctx, dlogits):
        model, x = ctx.model, ctx.x
        if ctx.saved is None:
            raise RuntimeError("forward ran without saving activations (no_grad); cannot backpropagate")
        N, T, t = ctx.dims
        spec = model._specs["net_token"]
        V, Vp = model.tokenizer.vocab_size, model.vocab_padded
        R = N * T
        dl = torch.zeros((R, Vp), dtype=model.dtype, device=model.device)
        dl[:, :V].copy_(dlogits.reshape(R, V))
        tmp = torch.empty_like(model._flat)
        G = model._stack_views("net_token", tmp)
        off, n, _ = model._offsets["lm_head.weight"]
        g_lm = tmp[off:off + n].view(V, spec.D)
        dh = torch.empty((R, spec.D), dtype=model.dtype, device=model.device)
        ops.gemm_nt(dl, model.lm_head.weight.data, dh, K=V, tb=True)
        ops.gemm_nt(dl, ctx.h, g_lm, K=R, ta=True, tb=True)
        dseq = engine.stack_backward(spec, model._W["net_token"], G, ctx.saved, dh,
                                     model.rope("net_token"), False)
        ctx.saved = ctx.h = None
        dhidden = torch.empty((N, spec.D), dtype=model.dtype, device=model.device)
        ops.copy_rows(dseq, T * spec.D, dhidden, spec.D, N, spec.D)
        acc = torch.zeros(G.embed.shape, dtype=torch.float32, device=model.device)
        if t > 0:
            ops.embed_scatter_bwd(x, t, dseq, T, 1, 1, acc, model.tokenizer.pad_id)
        ops.cast_from_f32(acc, G.embed, False)
        return (None, dhidden, None, *_grads_for(model, "net_token", tmp), g_lm)
