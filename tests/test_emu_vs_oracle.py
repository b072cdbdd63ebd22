"""The per-kernel GPU tests (tests/test_kernels_gpu.py) compare every C-ABI entry point with tests/emu_ops.py.  This file pins
emu_ops DIRECTLY to the oracle (oracle/midi_oracle.py, itself pinned to outputs of the reference in tests/test_oracle_golden.py):
each primitive's forward against the oracle's function of the same name, each backward against autograd through the oracle's
forward -- fp32 on CPU, so the only differences are summation order (tolerances 1e-5 relative)."""
import math

import numpy as np
import pytest
import torch

import emu_ops as emu


def rnd(shape, seed, scale=1.0):
    return scale * torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def close(a, b, rtol=2e-5, atol=2e-6, what=""):
    np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=rtol, atol=atol, err_msg=what)


def test_rmsnorm_forward_and_backward(orc):
    x, w, dy = rnd((37, 256), 1), 1.0 + 0.1 * rnd((256,), 2), rnd((37, 256), 3)
    y, rstd = torch.empty_like(x), torch.empty(37)
    emu.rmsnorm_fwd(x, w, y, rstd, 1e-6)
    close(y, orc.rmsnorm(x, w, 1e-6), what="rmsnorm forward")
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    orc.rmsnorm(xg, wg, 1e-6).backward(dy)
    dx, dw = torch.empty_like(x), torch.zeros(256)
    res = rnd((37, 256), 4)
    emu.rmsnorm_bwd(x, w, rstd, dy, res, dx, dw, False)
    close(dx, xg.grad + res, what="rmsnorm dx (+ residual gradient)")
    close(dw, wg.grad, rtol=1e-4, atol=1e-5, what="rmsnorm dw")


@pytest.mark.parametrize("hd,H", [(64, 3), (256, 2)])
def test_rope_matches_oracle_tables_and_rotation(orc, hd, H):
    B, S, pos0 = 2, 9, 5
    D = H * hd
    qkv = rnd((B * S, 3 * D), 11)
    cos, sin = orc.rope_tables(torch.arange(0, pos0 + S), hd)           # [pos, hd]; the kernel tables hold the first half
    got = emu.rope_(qkv.clone(), cos[:, : hd // 2].contiguous(), sin[:, : hd // 2].contiguous(), S, pos0, H, hd, +1)
    for part in range(2):
        x = qkv[:, part * D:(part + 1) * D].view(B, S, H, hd).transpose(1, 2)
        want = orc.rope(x, cos[pos0:pos0 + S], sin[pos0:pos0 + S]).transpose(1, 2).reshape(B * S, D)
        close(got[:, part * D:(part + 1) * D], want, what=f"rope part {part}")
    assert torch.equal(got[:, 2 * D:], qkv[:, 2 * D:])
    back = emu.rope_(got.clone(), cos[:, : hd // 2].contiguous(), sin[:, : hd // 2].contiguous(), S, pos0, H, hd, -1)
    close(back, qkv, rtol=1e-5, atol=1e-5, what="rotation back")


def test_event_attention_forward_and_backward(orc):
    B, S, H, hd = 2, 70, 3, 64
    D = H * hd
    qkv, do = rnd((B * S, 3 * D), 21), rnd((B * S, D), 22)
    Sp = emu.round_up(S, 64)
    o, lse = torch.empty((B * S, D)), torch.zeros(B * H * Sp)
    emu.attn_fwd(qkv, o, lse, B, S, H, hd ** -0.5)
    g = qkv.clone().requires_grad_(True)
    q, k, v = (g[:, i * D:(i + 1) * D].view(B, S, H, hd).transpose(1, 2) for i in range(3))
    want = orc.attention(q, k, v, causal=True).transpose(1, 2).reshape(B * S, D)
    close(o, want, what="attention forward")
    s = (q @ k.transpose(-1, -2)) * hd ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    close(lse.view(B, H, Sp)[:, :, :S], torch.logsumexp(s, -1), what="log-sum-exp")
    want.backward(do)
    dqkv = torch.empty_like(qkv)
    emu.attn_bwd(qkv, o, do, lse, dqkv, B, S, H, hd ** -0.5)
    close(dqkv, g.grad, rtol=1e-4, atol=1e-5, what="attention backward")


def test_token_attention_with_rope_forward_and_backward(orc):
    N, T, H, hd = 5, 8, 2, 256
    D = H * hd
    qkv, do = rnd((N * T, 3 * D), 31), rnd((N * T, D), 32)
    cos, sin = orc.rope_tables(torch.arange(T), hd)
    ch, sh = cos[:, : hd // 2].contiguous(), sin[:, : hd // 2].contiguous()
    o = torch.empty((N * T, D))
    emu.tokattn_fwd(qkv, o, N, T, H, hd ** -0.5, ch, sh)
    g = qkv.clone().requires_grad_(True)
    q, k, v = (g[:, i * D:(i + 1) * D].view(N, T, H, hd).transpose(1, 2) for i in range(3))
    want = orc.attention(orc.rope(q, cos, sin), orc.rope(k, cos, sin), v, causal=True).transpose(1, 2).reshape(N * T, D)
    close(o, want, what="token attention forward")
    want.backward(do)
    dqkv = torch.empty_like(qkv)
    emu.tokattn_bwd(qkv, do, dqkv, N, T, H, hd ** -0.5, ch, sh)
    close(dqkv, g.grad, rtol=1e-4, atol=1e-5, what="token attention backward (gradient of the UNROTATED q, k)")


def test_swiglu_mlp_forward_and_backward(orc):
    M, D, I = 19, 64, 96
    x, wg, wu, wd = rnd((M, D), 41), rnd((I, D), 42, 0.2), rnd((I, D), 43, 0.2), rnd((D, I), 44, 0.2)
    dy = rnd((M, D), 45)
    gu = torch.empty((M, 2 * I))
    a = torch.empty((M, I))
    emu.gemm_swiglu(x, torch.cat([wg, wu]), gu, a)
    out = emu.gemm_nt(a, wd, torch.empty((M, D)))
    xg = x.clone().requires_grad_(True)
    want = orc.mlp(xg, wg, wu, wd)
    close(out, want, rtol=1e-4, atol=1e-5, what="SwiGLU MLP forward")
    want.backward(dy)
    dgu = torch.empty((M, 2 * I))
    emu.gemm_dswiglu(dy, wd, gu, dgu)                                  # d a = dy Wd, then the SwiGLU backward
    dx = emu.gemm_nt(dgu, torch.cat([wg, wu]), torch.empty((M, D)), tb=True)
    close(dx, xg.grad, rtol=1e-4, atol=1e-5, what="SwiGLU MLP backward")


def test_cross_entropy_and_adamw_and_clip(orc):
    R, V, Vp = 33, 50, 56
    logits = rnd((R, Vp), 51)
    target = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(52))
    target[::5] = 0
    row_loss, dl = torch.empty(R), torch.empty((R, Vp))
    inv = torch.tensor([1.0 / float((target != 0).sum())])
    emu.cross_entropy(logits, V, target, row_loss, dl, inv, None, 0)
    lg = logits[:, :V].clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(lg, target, reduction="mean", ignore_index=0)
    assert abs(row_loss.sum().item() * inv.item() - want.item()) < 1e-5
    want.backward()
    close(dl[:, :V], lg.grad, rtol=1e-4, atol=1e-7, what="cross-entropy gradient")
    assert (dl[:, V:] == 0).all()
    # AdamW + clip against the oracle's step (itself held to torch.optim.AdamW by the golden tests)
    p, gr = rnd((77,), 61), rnd((77,), 62, 3.0)
    m, v = rnd((77,), 63, 0.1), rnd((77,), 64, 0.1).abs()
    coef, norm = orc.clip_coef([gr], 1.0)
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    orc.adamw_step(p2, gr * coef, m2, v2, 3, 2e-4, 0.01)
    ss, cf, nm = torch.zeros(1), torch.ones(1), torch.zeros(1)
    emu.sumsq(gr, torch.empty(1024), ss, False)
    emu.clip_coef(ss, 1.0, cf, nm)
    assert abs(nm.item() - norm.item()) < 1e-5 and abs(cf.item() - coef.item()) < 1e-6
    b1, b2 = 0.9, 0.99
    emu.adamw(p, gr, m, v, 2e-4, b1, b2, 1e-8, 0.01, 1 - b1 ** 3, 1 - b2 ** 3, cf)
    close(p, p2, rtol=1e-6, atol=1e-7, what="AdamW parameters")
    close(m, m2, what="AdamW m")
    close(v, v2, what="AdamW v")
