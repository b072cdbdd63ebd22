"""TEST-ONLY fake backend: torch-CPU stand-ins for every function of ``midi_model_amd.ops`` so the HOST logic
(kernel schedules of engine.py, the flat parameter/gradient layout, the fused training step, the decode loop,
the gradient reducer) can be exercised in the GPU-less container.  The product never imports this file; on a
GPU box the same tests run against the real HIP kernels (tests marked ``gpu``).

Each stand-in follows the C-ABI contract of include/midihip.h for its entry point (argument meaning, in-place
behaviour, buffer layouts such as lse[B,H,Sp]).
"""
from __future__ import annotations

import contextlib
import math

import torch
import torch.nn.functional as F


def set_option(name, value):
    pass


def get_option(name):
    return 1


def round_up(x, m):
    return (x + m - 1) // m * m


def gemm_nt(a, b, out, *, K=None, alpha=1.0, beta=0.0, res=None, splitk=0, ta=False, tb=False):
    if K is None:
        K = a.shape[0] if ta else a.shape[1]
    M, N = out.shape
    af = a[:K, :M].float().T if ta else a[:M, :K].float()
    bf = b[:K, :N].float().T if tb else b[:N, :K].float()
    acc = af @ bf.T
    r = alpha * acc
    if beta != 0.0:
        r = r + beta * (out if res is None else res).float()
    out.copy_(r.to(out.dtype))
    return out


SKINNY_PLAIN, SKINNY_GATEUP = 0, 1


def swiglu_fused_ok(x, I):
    return x.dtype == torch.bfloat16 and I % 128 == 0


def _scaled_product(x, w, out, rowscale):
    """out = round(rowscale[:, None] * (x @ w^T)): the fp32 product scaled BEFORE it is rounded (mh_gemm_*_scaled)"""
    out.copy_((rowscale.float()[:, None] * (x.float() @ w.float().T)).to(out.dtype))
    return out


def gemm_swiglu(x, wgu, gu, a, rowscale=None):
    if gu is None:  # forward-only form: gate|up is not kept
        gu = torch.empty((x.shape[0], wgu.shape[0]), dtype=x.dtype)
    if rowscale is not None:
        _scaled_product(x, wgu, gu, rowscale)
    else:
        gemm_nt(x, wgu, gu)
    return swiglu_fwd(gu, a)


def norm_fold_ok(x, D, hd, I):
    return rope_fused_ok(x, hd) and swiglu_fused_ok(x, I) and D % 64 == 0 and x.shape[0] % 4 == 0


def gemm_rowss(a, b, out, rowss, res=None):
    gemm_nt(a, b, out, beta=0.0 if res is None else 1.0, res=res)
    M, N = out.shape
    rowss.copy_(out.float().pow(2).view(M, N // 64, 64).sum(-1).T)
    return out


def row_rstd(rstd, D, eps, x=None, parts=None):
    ss = parts.sum(0) if parts is not None else x.float().pow(2).sum(-1)
    rstd.copy_(torch.rsqrt(ss / D + eps))
    return rstd


def rope_fused_ok(x, hd):
    return x.dtype == torch.bfloat16 and hd == 64


def gemm_rope(x, wqkv, qkv, table, S, pos0, hd, rowscale=None):
    """table: bf16 [npos, 96] = cos | -sin | +sin (RopeTable.fused)"""
    if rowscale is not None:
        _scaled_product(x, wqkv, qkv, rowscale)
    else:
        gemm_nt(x, wqkv, qkv)
    H = wqkv.shape[0] // (3 * hd)
    return rope_(qkv, table[:, :32].float(), table[:, 64:96].float(), S, pos0, H, hd, +1)


def dswiglu_ok(dx, I):
    return dx.dtype == torch.bfloat16 and I % 8 == 0


def gemm_dswiglu(dx, wd, gu, dgu, rowscale=None):
    da = torch.empty((dx.shape[0], wd.shape[1]), dtype=dx.dtype)
    if rowscale is not None:  # (mh_gemm_dswiglu_scaled: d a times rowscale[m] BEFORE it is rounded; SwiGLU' is linear in d a)
        da.copy_((rowscale.float()[:, None] * (dx.float() @ wd.float())).to(da.dtype))
    else:
        gemm_nt(dx, wd, da, tb=True)
    return swiglu_bwd(gu, da, dgu)


def gemm_nt_scaled(a, b, out, rowscale):
    return _scaled_product(a, b, out, rowscale)


def scale_cols(W, w, out):
    out.copy_((W.float() * w.float()[None, :]).to(out.dtype))
    return out


def scale_cols_jobs(triples):
    return list(triples)


def scale_cols_batched(jobs, K, like):
    for W, w, out in jobs:
        scale_cols(W, w, out)


def rmsnorm_bwd_folded(x, rstd, t, dres, dx):
    """dx = t - x (rstd^2 / D) rowdot(t, x) + dres (mh_rmsnorm_bwd_folded)"""
    xf, tf = x.float(), t.float()
    cf = rstd.float() ** 2 * (tf * xf).sum(-1) / x.shape[1]
    d = tf - xf * cf[:, None]
    if dres is not None:
        d = d + dres.float()
    dx.copy_(d.to(dx.dtype))
    return dx


def wgrad_folded(dz, x, dw, wnorm, W, dnorm, accumulate):
    """G' = dz^T x in fp32; dw (+)= G' * wnorm; dnorm (+)= colsum(G' * W)  (mh_gemm + mh_gemm_splitk_reduce_fold + mh_colsum)"""
    G = dz.float().T @ x.float()
    dw.copy_((G * wnorm.float()[None, :] + (dw.float() if accumulate else 0.0)).to(dw.dtype))
    dnorm.copy_(((G * W.float()).sum(0) + (dnorm.float() if accumulate else 0.0)).to(dnorm.dtype))


def attn_bwd_scaled_ok(qkv):
    return qkv.dtype == torch.bfloat16


def skinny_ok(x, K):
    return x.dtype == torch.bfloat16 and x.shape[0] <= 64 and K % 256 == 0


def gemm_skinny(a, w, out, *, mode=0, res=None, norm_eps=0.0, row_ids=None, res_ids=None):
    if row_ids is not None:
        a = a[row_ids]
    if res is not None and res_ids is not None:
        res = res[res_ids]
    r = a.float() @ w.float().T
    if norm_eps > 0:
        r = r * torch.rsqrt(a.float().pow(2).mean(-1, keepdim=True) + norm_eps)
    if mode == SKINNY_GATEUP:
        gu = r.to(a.dtype)
        swiglu_fwd(gu, out)
        return out
    if res is not None:
        r = r + res.float()
    out.copy_(r.to(out.dtype))
    return out


def transpose(x, out=None, pad_to=8):
    R, C = x.shape
    if out is None:
        out = torch.zeros((C, round_up(R, pad_to)), dtype=x.dtype)
    out[:C, :R] = x.T
    return out


def collate_windows(tokens_i16, win_start, win_len, out, pad_id):
    out.fill_(pad_id)
    for b in range(out.shape[0]):
        n = int(win_len[b])
        out[b, :n] = tokens_i16[int(win_start[b]): int(win_start[b]) + n].long()
    return out


def augment_piece_stats(tokens_i16, piece_off, tab, stats):
    """the whole-file facts of mh_augment_piece_stats (csrc/augment.hip), per piece: lowest / highest pitch among the notes off
    channel 9, and for every ORIGINAL track the bit mask of its notes' ORIGINAL channels"""
    t = [int(x) for x in tab]
    a = tokens_i16.long()
    for p in range(piece_off.numel() - 1):
        rows = a[int(piece_off[p]): int(piece_off[p + 1])]
        notes = rows[rows[:, 0] == t[1]]
        tr, ch, pitch = notes[:, t[7]] - t[26], notes[:, t[13]] - t[28], notes[:, t[19]] - t[30]
        off9 = pitch[ch != 9]
        stats[p, 0] = int(off9.min()) if off9.numel() else 128
        stats[p, 1] = int(off9.max()) if off9.numel() else -1
        stats[p, 2:] = 0
        for k in range(notes.shape[0]):
            stats[p, 2 + int(tr[k])] |= 1 << int(ch[k])
    return stats


def augment_collate_windows(tokens_i16, win_start, win_len, win_piece, shifts, stats, tab, out, pad_id):
    """mh_augment_collate_windows: row-local rules given the per-piece facts (the decomposition the device uses; the CPU tests
    hold it to the oracle's whole-file restatement and to the reference's goldens)"""
    t = [int(x) for x in tab]
    out.fill_(pad_id)

    def pymod(x, m):
        return ((x % m) + m) % m

    for b in range(out.shape[0]):
        n, s0 = int(win_len[b]), int(win_start[b])
        rows = tokens_i16[s0: s0 + n].long().clone()
        sh = [int(x) for x in shifts[b]]
        st = stats[int(win_piece[b])]
        lo, hi = int(st[0]), int(st[1])
        if not (hi >= 0 and (lo + sh[0] < 0 or hi + sh[0] > 127)):
            for r in range(n):
                v = [int(x) for x in rows[r]]
                if v[0] < 0 or v[0] not in t[1:7]:
                    continue
                e = t[1:7].index(v[0])
                tcol, ccol, c0, tr_new = t[7 + e], t[13 + e], -1, -1
                if tcol:
                    tr_new = pymod(v[tcol] - t[26] + sh[4], t[27])
                if ccol:
                    c0 = v[ccol] - t[28]
                    c = pymod(c0 + sh[5], t[29])
                    c = 9 if c0 == 9 else (pymod(9 + sh[5], t[29]) if c == 9 else c)
                    v[ccol] = t[28] + c
                if e == 0:
                    v[t[19]] += sh[0] if c0 != 9 else 0
                    v[t[20]] = t[31] + max(1, min(127, v[t[20]] - t[31] + sh[1]))
                elif e == 2:
                    if v[t[21]] - t[32] in (1, 2, 7, 11):
                        v[t[22]] = t[33] + max(1, min(127, v[t[22]] - t[33] + sh[2]))
                elif e == 3:
                    v[t[23]] = t[34] + max(1, min(t[35] - 1, v[t[23]] - t[34] + sh[3]))
                elif e == 5:
                    sf, mi = v[t[24]] - t[36] - 7, v[t[25]] - t[37]
                    k = pymod(pymod(sf * 7, 12) + sh[0], 12)
                    sf = (k * 7) % 12
                    if sf > 6 or (mi == 1 and sf >= 5):
                        sf -= 12
                    sf += 7
                    if 0 <= tr_new < 128 and int(st[2 + tr_new]) == 1 << 9:
                        sf = 7
                    v[t[24]] = t[36] + sf
                if tcol:
                    v[tcol] = t[26] + tr_new
                rows[r] = torch.tensor(v)
        out[b, :n] = rows
    return out


def embed_sum_fwd(tok, table, out):
    out.copy_(table[tok].float().sum(1).to(out.dtype))
    return out


def concat_tok_fwd(hidden, tok, table, out, T):
    out[:, 0] = hidden
    if T > 1:
        out[:, 1:] = table[tok[:, : T - 1]]
    return out


def embed_scatter_bwd(tok, T, dout, rows_per_m, jstride, j0, dtable_f32, pad_id):
    M = tok.shape[0]
    D = dtable_f32.shape[1]
    d = dout.reshape(-1, D).float()
    for j in range(T):
        ids = tok[:, j]
        rows = torch.arange(M) * rows_per_m + j * jstride + j0
        keep = ids != pad_id
        dtable_f32.index_add_(0, ids[keep], d[rows[keep]])


def token_segments(tok, V, row_mul=None, col_mul=1, add=0):
    if tok.dim() == 1:
        tok = tok.view(-1, 1)
    n_rows, n_cols = tok.shape
    if row_mul is None:
        row_mul = n_cols
    sorted_tok, order = torch.sort(tok.reshape(-1), stable=True)
    seg = torch.searchsorted(sorted_tok, torch.arange(V + 1, dtype=tok.dtype))
    src = torch.div(order, n_cols, rounding_mode="floor") * row_mul + (order % n_cols) * col_mul + add
    return src, seg.contiguous()


def embed_segment_bwd(src_rows, seg_start, dout, ld, dtable_f32, pad_id):
    V, D = dtable_f32.shape
    rows = torch.as_strided(dout, (int(src_rows.max()) + 1 if src_rows.numel() else 0, D), (ld, 1), dout.storage_offset()).float()
    ids = torch.repeat_interleave(torch.arange(V), seg_start[1:] - seg_start[:-1])
    keep = ids != pad_id
    dtable_f32.index_add_(0, ids[keep], rows[src_rows[keep]])


def cast_from_f32(src, dst, accumulate):
    flat = dst.view(-1)
    flat.copy_(((flat.float() if accumulate else 0) + src.view(-1)).to(dst.dtype))


def copy_rows(src, src_ld, dst, dst_ld, M, D, accumulate=False):
    s = torch.as_strided(src, (M, D), (src_ld, 1), src.storage_offset())
    d = torch.as_strided(dst, (M, D), (dst_ld, 1), dst.storage_offset())
    d.copy_((s.float() + (d.float() if accumulate else 0)).to(dst.dtype))


def rmsnorm_fwd(x, w, y, rstd, eps):
    xf = x.float()
    r = torch.rsqrt(xf.pow(2).mean(-1) + eps)
    if rstd is not None:
        rstd.copy_(r)
    y.copy_((w.float() * (xf * r[:, None]).to(x.dtype).float()).to(y.dtype))
    return y


def rmsnorm_bwd(x, w, rstd, dy, dres, dx, dw, accumulate):
    xf, g = x.float(), dy.float() * w.float()
    xh = xf * rstd[:, None]
    c = (g * xh).mean(-1, keepdim=True)
    d = rstd[:, None] * (g - xh * c)
    if dres is not None:
        d = d + dres.float()
    dwn = (dy.float() * xh.to(x.dtype).float()).sum(0)
    dx.copy_(d.to(dx.dtype))
    dw.copy_(((dw.float() if accumulate else 0) + dwn).to(dw.dtype))
    return dx


def rope_(qkv, cos_t, sin_t, S, pos0, H, hd, direction=1):
    M = qkv.shape[0]
    D = H * hd
    pos = pos0 + (torch.arange(M) % S)
    c = cos_t[pos].to(qkv.dtype).float()[:, None, :]
    s = direction * sin_t[pos].to(qkv.dtype).float()[:, None, :]
    for part in range(2):
        v = qkv[:, part * D:(part + 1) * D].float().view(M, H, hd)
        x1, x2 = v[..., : hd // 2], v[..., hd // 2:]
        o = torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1)
        qkv[:, part * D:(part + 1) * D] = o.reshape(M, D).to(qkv.dtype)
    return qkv


def _split(qkv, B, S, H, hd):
    D = H * hd
    q, k, v = (qkv[:, i * D:(i + 1) * D].float().view(B, S, H, hd).transpose(1, 2) for i in range(3))
    return q, k, v


def _causal_scores(q, k, scale):
    S = q.shape[-2]
    s = (q @ k.transpose(-1, -2)) * scale
    mask = torch.triu(torch.ones(S, S, dtype=torch.bool), 1)
    return s.masked_fill(mask, float("-inf"))


def attn_fwd(qkv, o, lse, B, S, H, scale):
    q, k, v = _split(qkv, B, S, H, 64)
    s = _causal_scores(q, k, scale)
    p = torch.softmax(s, -1)
    o.copy_((p @ v).transpose(1, 2).reshape(B * S, H * 64).to(o.dtype))
    Sp = round_up(S, 64)
    lse.view(B, H, Sp)[:, :, :S] = torch.logsumexp(s, -1)
    return o


def _attn_grads(q, k, v, do, scale):
    s = _causal_scores(q, k, scale)
    p = torch.softmax(s, -1)
    dv = p.transpose(-1, -2) @ do
    dp = do @ v.transpose(-1, -2)
    ds = p * (dp - (p * dp).sum(-1, keepdim=True)) * scale
    return ds @ k, ds.transpose(-1, -2) @ q, dv


def attn_bwd(qkv, o, dout, lse, dqkv, B, S, H, scale, cos_t=None, sin_t=None, rowscale=None):
    q, k, v = _split(qkv, B, S, H, 64)
    do = dout.float().view(B, S, H, 64).transpose(1, 2)
    dq, dk, dv = _attn_grads(q, k, v, do, scale)
    D = H * 64
    for i, t in enumerate((dq, dk, dv)):
        t = t.transpose(1, 2).reshape(B * S, D)
        if rowscale is not None:  # (the kernels scale the fp32 accumulators before the store's rounding)
            t = t * rowscale.float()[:, None]
        dqkv[:, i * D:(i + 1) * D] = t.to(dqkv.dtype)
    if cos_t is not None:
        rope_(dqkv, cos_t, sin_t, S, 0, H, 64, -1)
    return dqkv


def tokattn_fwd(qkv, o, N, T, H, scale, cos_t=None, sin_t=None):
    if cos_t is not None:
        qkv = rope_(qkv.clone(), cos_t, sin_t, T, 0, H, 256, +1)
    q, k, v = _split(qkv, N, T, H, 256)
    p = torch.softmax(_causal_scores(q, k, scale), -1)
    o.copy_((p @ v).transpose(1, 2).reshape(N * T, H * 256).to(o.dtype))
    return o


def tokattn_bwd(qkv, dout, dqkv, N, T, H, scale, cos_t=None, sin_t=None, rowscale=None):
    if cos_t is not None:
        qkv = rope_(qkv.clone(), cos_t, sin_t, T, 0, H, 256, +1)
    q, k, v = _split(qkv, N, T, H, 256)
    do = dout.float().view(N, T, H, 256).transpose(1, 2)
    dq, dk, dv = _attn_grads(q, k, v, do, scale)
    D = H * 256
    for i, t in enumerate((dq, dk, dv)):
        t = t.transpose(1, 2).reshape(N * T, D)
        if rowscale is not None:  # (a row scale commutes with the rotation back)
            t = t * rowscale.float()[:, None]
        dqkv[:, i * D:(i + 1) * D] = t.to(dqkv.dtype)
    if cos_t is not None:
        rope_(dqkv, cos_t, sin_t, T, 0, H, 256, -1)
    return dqkv


def swiglu_fwd(gu, a):
    I = gu.shape[1] // 2
    g, u = gu[:, :I].float(), gu[:, I:].float()
    a.copy_((F.silu(g).to(gu.dtype).float() * u).to(a.dtype))
    return a


def swiglu_bwd(gu, da, dgu):
    I = gu.shape[1] // 2
    g, u, d = gu[:, :I].float(), gu[:, I:].float(), da.float()
    sig = torch.sigmoid(g)
    dgu[:, :I] = (d * u * (sig * (1 + g * (1 - sig)))).to(dgu.dtype)
    dgu[:, I:] = (d * g * sig).to(dgu.dtype)
    return dgu


def cross_entropy(logits, V, target, row_loss, dlogits=None, scale_dev=None, argmax_out=None, ignore=0):
    lg = logits[:, :V].float()
    lse = torch.logsumexp(lg, -1)
    keep = target != ignore
    tl = lg.gather(1, target[:, None]).squeeze(1)
    row_loss.copy_(torch.where(keep, lse - tl, torch.zeros_like(lse)))
    if argmax_out is not None:
        argmax_out.copy_(lg.argmax(-1))
    if dlogits is not None:
        g = torch.softmax(lg, -1)
        g[torch.arange(g.shape[0]), target] -= 1
        g = g * (scale_dev[0] if scale_dev is not None else 1.0)
        g[~keep] = 0
        dlogits.zero_()
        dlogits[:, :V] = g.to(dlogits.dtype)


def sum_f32(x, out):
    out[0] = x.sum()


def count_valid(target, ignore, count, inv):
    c = (target != ignore).sum().float()
    count[0] = c
    inv[0] = 1.0 / max(c.item(), 1.0)


def sumsq(g, partial1024, out, accumulate):
    s = g.float().pow(2).sum()
    out[0] = (out[0] if accumulate else 0) + s


def clip_coef(sumsq_t, max_norm, coef, norm):
    n = sumsq_t[0].sqrt()
    norm[0] = n
    coef[0] = min(1.0, max_norm / (n.item() + 1e-6))


def adamw(p, g, m, v, lr, b1, b2, eps, wd, bc1, bc2, coef_dev):
    T = p.dtype
    r = lambda t: t.to(T).float()
    gr = r(g.float() * (coef_dev[0] if coef_dev is not None else 1.0))
    pe = r(p.float() * (1 - lr * wd))
    me = r(m.float() + (gr - m.float()) * (1 - b1))
    ve = r(r(v.float() * b2) + (1 - b2) * gr * gr)
    den = r(r(r(ve.sqrt()) / math.sqrt(bc2)) + eps)
    p.copy_((pe - (lr / bc1) * (me / den)).to(T))
    m.copy_(me.to(T))
    v.copy_(ve.to(T))


def kv_append(qkv, cos_t, sin_t, kc, vc, B, H, hd, Lmax, pos, pos_dev=None):
    if pos_dev is not None:
        pos = int(pos_dev.item())
    rope_(qkv, cos_t, sin_t, 1, pos, H, hd, 1)
    D = H * hd
    kc[:, :, pos] = qkv[:, D:2 * D].view(B, H, hd)
    vc[:, :, pos] = qkv[:, 2 * D:].view(B, H, hd)


def attn_decode(qkv, kc, vc, o, B, H, hd, Lmax, length, scale, pos_dev=None):
    if pos_dev is not None:
        length = int(pos_dev.item()) + 1
    q = qkv[:, : H * hd].float().view(B, H, 1, hd)
    k, v = kc[:, :, :length].float(), vc[:, :, :length].float()
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1)
    o.copy_((p @ v).reshape(B, H * hd).to(o.dtype))
    return o


def attn_decode_append(qkv, cos_t, sin_t, kc, vc, o, B, H, hd, Lmax, pos, scale, pos_dev=None):
    if pos_dev is not None:
        pos = int(pos_dev.item())
    tmp = qkv.clone()
    kv_append(tmp, cos_t, sin_t, kc, vc, B, H, hd, Lmax, pos)
    return attn_decode(tmp, kc, vc, o, B, H, hd, Lmax, pos + 1, scale)


def kv_store_rows(qkv, kc, vc, B, S, H, hd, Lmax, pos0):
    D = H * hd
    kc[:, :, pos0:pos0 + S] = qkv[:, D:2 * D].view(B, S, H, hd).transpose(1, 2)
    vc[:, :, pos0:pos0 + S] = qkv[:, 2 * D:].view(B, S, H, hd).transpose(1, 2)


def kv_gather_rows(kc, vc, qkv, B, n, Stot, H, hd, Lmax):
    D = H * hd
    rows = qkv.view(B, Stot, 3 * D)
    rows[:, :n, :D] = 0
    rows[:, :n, D:2 * D] = kc[:, :, :n].transpose(1, 2).reshape(B, n, D)
    rows[:, :n, 2 * D:] = vc[:, :, :n].transpose(1, 2).reshape(B, n, D)


def attn_fwd_tail(qkv, o, lse, B, S, H, scale, q_start):
    """only rows >= q_start are meaningful (the device also writes the rest of the first query tile: not modelled)"""
    full = torch.empty_like(o)
    attn_fwd(qkv, full, lse, B, S, H, scale)
    D = o.shape[1]
    o.view(B, S, D)[:, q_start:] = full.view(B, S, D)[:, q_start:]
    return o


def kv_store_prefill(qkv, kc, vc, B, S, H, hd, Lmax):
    D = H * hd
    kc[:, :, :S] = qkv[:, D:2 * D].view(B, S, H, hd).transpose(1, 2)
    vc[:, :, :S] = qkv[:, 2 * D:].view(B, S, H, hd).transpose(1, 2)


SAMPLE_MAX_K = 64


SAMPLE_MAX_RANGE = 2048


def mask_spans(first_mask, lo_tab, hi_tab):
    nz = first_mask.nonzero().flatten()
    return (int(nz.min()), int(nz.max()) + 1), [int(x) for x in (hi_tab - lo_tab).max(dim=0).values.tolist()]


def sample_top_p_k(logits, first_mask, lo_tab, hi_tab, ev, pos, q, out, V, temp, top_p, top_k, out_b=None, out_c=None,
                   first_span=(0, 0), max_range=0, ban_mask=None, fill_rest=0, fill_id=0):
    B = logits.shape[0]
    if pos == 0:
        lo = hi = torch.full((B,), -1, dtype=torch.int32)
    else:
        lo, hi = lo_tab[ev, pos], hi_tab[ev, pos]
    probs = torch.empty((B, V), dtype=torch.float32)
    masked_softmax(logits, lo, hi, first_mask, probs, V, temp)
    if ban_mask is not None:
        probs = probs * (ban_mask == 0)[None, :]
    ps, pi = torch.sort(probs, dim=-1, descending=True, stable=True)
    cum = torch.cumsum(ps, -1)
    ps[cum - ps > top_p] = 0.0
    ps[:, top_k:] = 0.0
    ps = ps / ps.sum(-1, keepdim=True)
    j = torch.argmax(ps / q, -1)
    ids = pi.gather(-1, j[:, None])[:, 0]
    out.copy_(ids)
    if fill_rest:  # `out` is column `pos` of a [B, T] buffer: the following columns get the pad id
        flat = out.as_strided((out.shape[0], fill_rest + 1), (out.stride(0), 1), out.storage_offset())
        flat[:, 1:] = fill_id
    for t in (out_b, out_c):
        if t is not None:
            t.copy_(ids)
    return out


def masked_softmax(logits, lo, hi, first_mask, probs, V, temp):
    T = logits.dtype
    p = torch.softmax((logits[:, :V].float() / temp).to(T).float(), -1)
    ids = torch.arange(V)[None, :]
    rng = (ids >= lo[:, None]) & (ids < hi[:, None])
    mask = torch.where(lo[:, None] < 0, first_mask[None, :].bool().expand_as(rng), rng)
    probs.copy_(p * mask)
    return probs


_NAMES = [n for n, f in list(globals().items()) if callable(f) and not n.startswith("_") and n not in
          ("contextlib", "math", "torch", "F", "install", "round_up")]


@contextlib.contextmanager
def install():
    """Patch midi_model_amd.ops with the CPU stand-ins and lift the device check of MIDIModel."""
    import midi_model_amd.ops as real
    import midi_model_amd.model as model
    saved = {n: getattr(real, n) for n in _NAMES if hasattr(real, n)}
    missing = [n for n in dir(real) if not n.startswith("_") and callable(getattr(real, n)) and n not in saved
               and n not in ("lib", "dt", "round_up", "Optional", "ab_library")]
    assert not missing, f"emulator lacks stand-ins for {missing}"
    req = model.MIDIModel._require_gpu
    try:
        for n in saved:
            setattr(real, n, globals()[n])
        model.MIDIModel._require_gpu = lambda self: None
        yield
    finally:
        for n, f in saved.items():
            setattr(real, n, f)
        model.MIDIModel._require_gpu = req
