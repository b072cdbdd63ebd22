"""Forward / backward / decode schedules of one LLaMA stack (event-level `net` or token-level `net_token`)
as explicit sequences of C-ABI kernel launches — no autograd graph, no tracing compiler.

Block wiring follows TF:models/llama/modeling_llama.py:295-324 (pre-norm residual layer) and :367-417
(model: layers + final norm); see SURVEY.md §8 a5-a9.  Weights of a layer are views into the model's flat
parameter buffer, with q|k|v and gate|up stored contiguously so each pair is ONE projection GEMM.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import os

import torch

from . import ops


@dataclass
class StackSpec:
    name: str            # "net" | "net_token"
    D: int
    H: int
    I: int
    L: int
    eps: float
    theta: float
    kind: str            # "event" (causal flash attention, head_dim 64) | "token" (<=8-token sequences, head_dim 256)

    @property
    def hd(self) -> int:
        return self.D // self.H

    @property
    def scale(self) -> float:
        return self.hd ** -0.5


@dataclass
class LayerTensors:
    wqkv: torch.Tensor = None   # [3D, D]
    wo: torch.Tensor = None     # [D, D]
    wgu: torch.Tensor = None    # [2I, D]
    wd: torch.Tensor = None     # [D, I]
    n1: torch.Tensor = None     # [D]
    n2: torch.Tensor = None     # [D]


@dataclass
class StackTensors:
    """One of: weights, transposed weights ([in,out] copies for dgrad), gradients."""
    embed: torch.Tensor = None  # [V, D]
    layers: List[LayerTensors] = field(default_factory=list)
    norm: torch.Tensor = None   # [D]


class RopeTable:
    """fp32 cos/sin of pos * theta^(-2i/hd), [npos, hd/2] (TF:models/llama/modeling_llama.py:113-127)."""

    def __init__(self, hd: int, theta: float, device, npos: int = 0):
        self.hd, self.theta, self.device = hd, theta, device
        self.cos = self.sin = None
        self._fused = None
        self.n = 0
        if npos:
            self.ensure(npos)

    def ensure(self, npos: int):
        if npos <= self.n:
            return
        npos = max(npos, 2 * self.n, 64)
        inv_freq = 1.0 / (self.theta ** (torch.arange(0, self.hd, 2, dtype=torch.int64).float() / self.hd))
        ang = torch.arange(npos).float()[:, None] * inv_freq[None, :]
        self.cos = ang.cos().contiguous().to(self.device)
        self.sin = ang.sin().contiguous().to(self.device)
        self.n = npos
        self._fused = None

    def fused(self) -> torch.Tensor:
        """bf16 [npos, 96] = cos | -sin | +sin (head_dim 64): the table of mh_gemm_rope's epilogue"""
        if self._fused is None or self._fused.shape[0] != self.n:
            c, s = self.cos.to(torch.bfloat16), self.sin.to(torch.bfloat16)
            self._fused = torch.cat([c, -s, s], dim=1).contiguous()
        return self._fused


def _empty(shape, like: torch.Tensor, dtype=None):
    return torch.empty(shape, dtype=dtype or like.dtype, device=like.device)


def _check_heads(spec: StackSpec):
    want = 64 if spec.kind == "event" else 256
    if spec.hd != want:
        raise NotImplementedError(
            f"{spec.name}: head_dim {spec.hd} is not implemented by the HIP attention kernels "
            f"(event-level net needs 64, token-level net needs 256)")


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, accumulate: bool):
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]: the contraction runs over the M rows, i.e. both operands are
    contraction-major as they lie in HBM; the GEMM reads them in that form (split-K over M)."""
    ops.gemm_nt(dy, x, dw, K=dy.shape[0], ta=True, tb=True, beta=1.0 if accumulate else 0.0)


# --------------------------------------------------------------------------------------------------
# training / prefill forward
# --------------------------------------------------------------------------------------------------
def layer_forward(spec: StackSpec, lw: LayerTensors, x: torch.Tensor, nseq: int, slen: int, rope: RopeTable,
                  kv_out: Optional[list] = None, save: bool = True, lean: bool = False):
    """One pre-norm LLaMA block (LlamaDecoderLayer.forward, TF:models/llama/modeling_llama.py:295-324) on x [nseq*slen, D]:
    RMSNorm -> q|k|v projection -> RoPE -> causal attention -> o projection + residual -> RMSNorm -> gate|up projection with
    SwiGLU epilogue -> down projection + residual.  8 launches for the event-level stack in bf16 (bench.py --mode block times
    exactly this function).  ``save=False`` is the forward-only form (prompt prefill, validation): gate|up is never written,
    only the activation.  ``lean`` (with save): the SwiGLU activation ``a`` is not kept -- the backward recomputes it from the
    stored gate|up with mh_swiglu_fwd, bit for bit (the fused epilogue and that kernel share their roundings) -- which takes
    I of the 8 D + 3 I saved elements per row off the activation memory (the 2x-hidden large shape at 16 x 4096 per GPU: 39 GB).
    Returns (block output, tensors the backward needs | None)."""
    M, D = x.shape
    H, I = spec.H, spec.I
    h1 = _empty((M, D), x)
    rstd1 = _empty((M,), x, torch.float32)
    ops.rmsnorm_fwd(x, lw.n1, h1, rstd1, spec.eps)
    qkv = _empty((M, 3 * D), x)
    # token-level stack: RoPE is applied inside the attention kernels (q,k of a (sequence, head) are in registers
    # there anyway), so qkv stays unrotated -- except for a prefill, whose K rows go to the cache rotated.
    # event-level stack (heads of 64): RoPE rides on the q|k|v projection's epilogue.
    rope_in_attn = spec.kind != "event" and kv_out is None
    if not rope_in_attn and ops.rope_fused_ok(h1, spec.hd):
        ops.gemm_rope(h1, lw.wqkv, qkv, rope.fused(), slen, 0, spec.hd)
    else:
        ops.gemm_nt(h1, lw.wqkv, qkv)
        if not rope_in_attn:
            ops.rope_(qkv, rope.cos, rope.sin, slen, 0, H, spec.hd, +1)
    o = _empty((M, D), x)
    lse = None
    if spec.kind == "event":
        lse = _empty((nseq * H * ops.round_up(slen, 64),), x, torch.float32)
        ops.attn_fwd(qkv, o, lse, nseq, slen, H, spec.scale)
    elif rope_in_attn:
        ops.tokattn_fwd(qkv, o, nseq, slen, H, spec.scale, rope.cos, rope.sin)
    else:
        ops.tokattn_fwd(qkv, o, nseq, slen, H, spec.scale)
    if kv_out is not None:
        kv_out.append(qkv)
    x2 = _empty((M, D), x)
    ops.gemm_nt(o, lw.wo, x2, beta=1.0, res=x)
    h2 = _empty((M, D), x)
    rstd2 = _empty((M,), x, torch.float32)
    ops.rmsnorm_fwd(x2, lw.n2, h2, rstd2, spec.eps)
    a = _empty((M, I), x)
    gu = None
    if ops.swiglu_fused_ok(h2, I):                  # gate|up projection with SwiGLU as its epilogue
        if save:
            gu = _empty((M, 2 * I), x)
        ops.gemm_swiglu(h2, lw.wgu, gu, a)
    else:
        gu = _empty((M, 2 * I), x)
        ops.gemm_nt(h2, lw.wgu, gu)
        ops.swiglu_fwd(gu, a)
    x3 = _empty((M, D), x)
    ops.gemm_nt(a, lw.wd, x3, beta=1.0, res=x2)
    return x3, ((x, rstd1, h1, qkv, o, lse, x2, rstd2, h2, gu, None if lean else a) if save else None)


# The forward-only event-level block with both RMSNorms folded around its projections (r05): below these row counts the two
# extra ~3 us statistics launches (and, without pre-folded weights, the fold itself: ~20 us per layer of elementwise work) cost more
# than the two norm passes they replace (2 x 44 us at 65536 rows, proportional to the rows).
FOLD_MIN_ROWS_PREFOLDED = 8192
FOLD_MIN_ROWS_ON_THE_FLY = 131072   # (12 layers: the fold ~0.7 ms of elementwise launches against ~14 us saved per layer and 8192 rows)


def layer_forward_folded(spec: StackSpec, lw: LayerTensors, fold, x: torch.Tensor, nseq: int, slen: int, rope: RopeTable,
                         parts_in: Optional[torch.Tensor], kv_out: Optional[list] = None):
    """layer_forward(save=False) without its two RMSNorm passes (LlamaDecoderLayer.forward, TF:models/llama/modeling_llama.py:295-324;
    LlamaRMSNorm :62-67): ``fold`` = (wqkv * n1, wgu * n2) from fold_norm_weights, so norm(x) W^T = rstd (.) (x W'^T) and the
    normalised activations h1 / h2 are never written or read.  The row statistics come out of the PRODUCING projections: the
    o and down projections (mh_gemm_rowss) leave per-64-column sums of squares of the rows they store (``parts`` [D/64, M]),
    mh_row_rstd turns them into rstd (one tiny launch), the q|k|v + RoPE and gate|up + SwiGLU projections apply it to their
    fp32 accumulators before their own epilogue arithmetic.  7 launches, none of them a pass over the residual stream.
    ``parts_in``: the statistics of ``x`` left by the previous block's down projection (None: computed from x itself).
    Returns (block output, its parts)."""
    M, D = x.shape
    H, I = spec.H, spec.I
    wq_n, wgu_n = fold
    rstd = _empty((M,), x, torch.float32)
    if parts_in is not None:
        ops.row_rstd(rstd, D, spec.eps, parts=parts_in)
    else:
        ops.row_rstd(rstd, D, spec.eps, x=x)
    qkv = _empty((M, 3 * D), x)
    ops.gemm_rope(x, wq_n, qkv, rope.fused(), slen, 0, spec.hd, rowscale=rstd)
    o = _empty((M, D), x)
    lse = _empty((nseq * H * ops.round_up(slen, 64),), x, torch.float32)
    ops.attn_fwd(qkv, o, lse, nseq, slen, H, spec.scale)
    if kv_out is not None:
        kv_out.append(qkv)
    parts = _empty((D // 64, M), x, torch.float32)
    x2 = _empty((M, D), x)
    ops.gemm_rowss(o, lw.wo, x2, parts, res=x)
    ops.row_rstd(rstd, D, spec.eps, parts=parts)
    a = _empty((M, I), x)
    ops.gemm_swiglu(x2, wgu_n, None, a, rowscale=rstd)
    x3 = _empty((M, D), x)
    ops.gemm_rowss(a, lw.wd, x3, parts, res=x2)
    return x3, parts


def train_fold_ok(spec: StackSpec, x: torch.Tensor) -> bool:
    """whether the TRAINING forward / backward of stack ``spec`` over the rows ``x`` can run with its RMSNorms folded around
    the projections (layer_forward_train_folded / layer_backward_folded): bf16 on the production GEMM with its K-step-64 loops,
    the fused SwiGLU epilogues in both directions, whole 64-column statistics chunks, 4-row groups, and -- event-level stack --
    the RoPE epilogue and the one-call attention backward"""
    ok = (x.dtype == torch.bfloat16 and ops.swiglu_fused_ok(x, spec.I) and ops.dswiglu_ok(x, spec.I) and spec.D % 64 == 0
          and x.shape[0] % 4 == 0 and ops.get_option("gemm_k64") == 1 and os.environ.get("MH_NORM_FOLD_TRAIN", "1") != "0")
    if ok and spec.kind == "event":
        ok = ops.rope_fused_ok(x, spec.hd) and ops.attn_bwd_scaled_ok(x)
    return ok


def layer_forward_train_folded(spec: StackSpec, lw: LayerTensors, fold, x: torch.Tensor, nseq: int, slen: int, rope: RopeTable,
                               parts_in: Optional[torch.Tensor], lean: bool = False):
    """layer_forward(save=True) with both RMSNorms folded around the projections (r06; LlamaDecoderLayer.forward,
    TF:models/llama/modeling_llama.py:295-324, LlamaRMSNorm :62-67): the q|k|v and gate|up projections read the residual stream
    itself against ``fold`` = (wqkv * n1, wgu * n2) and scale their rows by rstd, whose statistics the PRODUCING projections (o,
    down: mh_gemm_rowss) leave behind.  The normalised activations h1 / h2 are never written, read or kept: two passes over the
    residual stream and a fifth of the saved activations less per layer.  Both stacks (token-level: RoPE stays inside the attention
    kernel, the q|k|v projection is the plain GEMM with a row scale).  Returns (output, its statistics, what the backward needs)."""
    M, D = x.shape
    H, I = spec.H, spec.I
    wq_n, wgu_n = fold
    rstd1 = _empty((M,), x, torch.float32)
    if parts_in is not None:
        ops.row_rstd(rstd1, D, spec.eps, parts=parts_in)
    else:
        ops.row_rstd(rstd1, D, spec.eps, x=x)
    qkv = _empty((M, 3 * D), x)
    o = _empty((M, D), x)
    lse = None
    if spec.kind == "event":
        ops.gemm_rope(x, wq_n, qkv, rope.fused(), slen, 0, spec.hd, rowscale=rstd1)
        lse = _empty((nseq * H * ops.round_up(slen, 64),), x, torch.float32)
        ops.attn_fwd(qkv, o, lse, nseq, slen, H, spec.scale)
    else:
        ops.gemm_nt_scaled(x, wq_n, qkv, rstd1)
        ops.tokattn_fwd(qkv, o, nseq, slen, H, spec.scale, rope.cos, rope.sin)
    parts = _empty((D // 64, M), x, torch.float32)
    x2 = _empty((M, D), x)
    ops.gemm_rowss(o, lw.wo, x2, parts, res=x)
    rstd2 = _empty((M,), x, torch.float32)
    ops.row_rstd(rstd2, D, spec.eps, parts=parts)
    gu = _empty((M, 2 * I), x)
    a = _empty((M, I), x)
    ops.gemm_swiglu(x2, wgu_n, gu, a, rowscale=rstd2)
    parts3 = _empty((D // 64, M), x, torch.float32)
    x3 = _empty((M, D), x)
    ops.gemm_rowss(a, lw.wd, x3, parts3, res=x2)
    return x3, parts3, (x, rstd1, qkv, o, lse, x2, rstd2, gu, None if lean else a)


def layer_backward_folded(spec: StackSpec, lw: LayerTensors, lg: LayerTensors, fold, keep, dx: torch.Tensor, nseq: int, slen: int,
                          rope: RopeTable, accumulate: bool) -> torch.Tensor:
    """The backward of layer_forward_train_folded.  With z = x W'^T, y = rstd (.) z: the producers of d y store d z = rstd (.) d y
    (the SwiGLU-backward epilogue, the attention backward's stores), t = d z W' is the dgrad on the folded weights,
    dx = t - x (rstd^2 / D) rowdot(t, x) + dres the norm's backward without its weight, and the weight gradient G' = d z^T x
    turns into dW = G' (.) w and dw = colsum(G' (.) W) inside its split-K reduction (ops.wgrad_folded)."""
    x, rstd1, qkv, o, lse, x2, rstd2, gu, a = keep
    M, D = dx.shape
    H, I = spec.H, spec.I
    wq_n, wgu_n = fold
    # ---- MLP ----
    if a is None:
        a = _empty((M, I), dx)
        ops.swiglu_fwd(gu, a)
    dz2 = _empty((M, 2 * I), dx)
    ops.gemm_dswiglu(dx, lw.wd, gu, dz2, rowscale=rstd2)     # rstd2 (.) SwiGLU'(gate|up) (dx @ wd)
    linear_wgrad(dx, a, lg.wd, accumulate)
    del a
    t2 = _empty((M, D), dx)
    ops.gemm_nt(dz2, wgu_n, t2, tb=True)                     # t2 = d z2 @ W'gu
    ops.wgrad_folded(dz2, x2, lg.wgu, lw.n2, lw.wgu, lg.n2, accumulate)
    del dz2
    dx2 = _empty((M, D), dx)
    ops.rmsnorm_bwd_folded(x2, rstd2, t2, dx, dx2)
    # ---- attention ----
    do = t2                                                  # reuse
    ops.gemm_nt(dx2, lw.wo, do, tb=True)
    linear_wgrad(dx2, o, lg.wo, accumulate)
    dz1 = _empty((M, 3 * D), dx)
    if spec.kind == "event":
        ops.attn_bwd(qkv, o, do, lse, dz1, nseq, slen, H, spec.scale, rope.cos, rope.sin, rowscale=rstd1)
    else:
        ops.tokattn_bwd(qkv, do, dz1, nseq, slen, H, spec.scale, rope.cos, rope.sin, rowscale=rstd1)
    t1 = do
    ops.gemm_nt(dz1, wq_n, t1, tb=True)                      # t1 = d z1 @ W'qkv
    ops.wgrad_folded(dz1, x, lg.wqkv, lw.n1, lw.wqkv, lg.n1, accumulate)
    del dz1
    ops.rmsnorm_bwd_folded(x, rstd1, t1, dx2, dx)
    return dx


def stack_forward(spec: StackSpec, W: StackTensors, x: torch.Tensor, nseq: int, slen: int, rope: RopeTable,
                  save: bool, kv_out: Optional[list] = None, lean: bool = False, folded=None):
    """x [nseq*slen, D] (inputs_embeds) -> last_hidden_state [nseq*slen, D].
    save=True keeps what the backward needs (``lean``: minus the SwiGLU activations, recomputed in the backward);
    kv_out (prefill) receives each layer's post-RoPE qkv.  Forward-only passes of the event-level stack over enough rows run
    the folded-norm blocks (layer_forward_folded); ``folded`` = fold_norm_weights(W) kept current by the caller (a decode
    session's), otherwise the fold is made here, from the live weights, per call."""
    _check_heads(spec)
    M, D = x.shape
    assert M == nseq * slen
    rope.ensure(slen)
    if (not save and spec.kind == "event" and M >= (FOLD_MIN_ROWS_PREFOLDED if folded is not None else FOLD_MIN_ROWS_ON_THE_FLY)
            and ops.norm_fold_ok(x, D, spec.hd, spec.I)):
        if folded is None:
            folded = fold_norm_weights(W)
        parts = None
        for lw, fold in zip(W.layers, folded):
            x, parts = layer_forward_folded(spec, lw, fold, x, nseq, slen, rope, parts, kv_out)
        y = _empty((M, D), x)
        ops.rmsnorm_fwd(x, W.norm, y, None, spec.eps)
        return y, None
    saved = []
    if save and folded is not None and kv_out is None and train_fold_ok(spec, x):
        # the TRAINING forward with folded RMSNorms (r06): ``folded`` = the weights of this step (the caller re-derives them after
        # every update: MIDIModel.folded_weights); the backward finds them in the context
        parts = None
        for lw, fold in zip(W.layers, folded):
            x, parts, keep = layer_forward_train_folded(spec, lw, fold, x, nseq, slen, rope, parts, lean)
            saved.append(keep)
        y = _empty((M, D), x)
        rstdf = _empty((M,), x, torch.float32)
        ops.rmsnorm_fwd(x, W.norm, y, rstdf, spec.eps)
        return y, (saved, x, rstdf, nseq, slen, folded)
    for lw in W.layers:
        x3, keep = layer_forward(spec, lw, x, nseq, slen, rope, kv_out, save, lean and save)
        if save:
            saved.append(keep)
        x = x3
    y = _empty((M, D), x)
    rstdf = _empty((M,), x, torch.float32)
    ops.rmsnorm_fwd(x, W.norm, y, rstdf, spec.eps)
    return y, ((saved, x, rstdf, nseq, slen) if save else None)


def stack_backward(spec: StackSpec, W: StackTensors, G: StackTensors, ctx, dy: torch.Tensor,
                   rope: RopeTable, accumulate: bool, on_layer_done: Optional[Callable[[int], None]] = None):
    """dy = d loss / d last_hidden_state  ->  d loss / d inputs_embeds; parameter gradients go to G
    (overwritten, or added to when `accumulate`).  `on_layer_done(i)` fires when layer i's gradients are
    final (layers finish in reverse order) — the data-parallel reducer hangs its bucket launches on it."""
    folded = ctx[5] if len(ctx) > 5 else None
    saved, x_last, rstdf, nseq, slen = ctx[:5]
    M, D = dy.shape
    H, I = spec.H, spec.I
    dx = _empty((M, D), dy)
    ops.rmsnorm_bwd(x_last, W.norm, rstdf, dy, None, dx, G.norm, accumulate)
    if folded is not None:  # the forward ran layer_forward_train_folded
        for li in range(len(W.layers) - 1, -1, -1):
            dx = layer_backward_folded(spec, W.layers[li], G.layers[li], folded[li], saved[li], dx, nseq, slen, rope, accumulate)
            saved[li] = None
            if on_layer_done is not None:
                on_layer_done(li)
        return dx
    for li in range(len(W.layers) - 1, -1, -1):
        lw, lg = W.layers[li], G.layers[li]
        x, rstd1, h1, qkv, o, lse, x2, rstd2, h2, gu, a = saved[li]
        # ---- MLP ----
        if a is None:                                   # lean forward: a = round(silu(gate)) * up again, from the stored gate|up
            a = _empty((M, I), dy)
            ops.swiglu_fwd(gu, a)
        dgu = _empty((M, 2 * I), dy)
        if ops.dswiglu_ok(dx, I):                       # d a = dx @ wd with the SwiGLU backward as its epilogue
            ops.gemm_dswiglu(dx, lw.wd, gu, dgu)
        else:
            da = _empty((M, I), dy)
            ops.gemm_nt(dx, lw.wd, da, tb=True)         # d a = dx @ wd      (wd [D, I] read contraction-major)
            ops.swiglu_bwd(gu, da, dgu)
            del da
        linear_wgrad(dx, a, lg.wd, accumulate)
        del a
        dh2 = _empty((M, D), dy)
        ops.gemm_nt(dgu, lw.wgu, dh2, tb=True)          # d h2 = dgu @ wgu
        linear_wgrad(dgu, h2, lg.wgu, accumulate)
        del dgu
        dx2 = _empty((M, D), dy)
        ops.rmsnorm_bwd(x2, lw.n2, rstd2, dh2, dx, dx2, lg.n2, accumulate)
        # ---- attention ----
        do = dh2                                        # reuse
        ops.gemm_nt(dx2, lw.wo, do, tb=True)            # d o = dx2 @ wo
        linear_wgrad(dx2, o, lg.wo, accumulate)
        dqkv = _empty((M, 3 * D), dy)
        if spec.kind == "event":
            ops.attn_bwd(qkv, o, do, lse, dqkv, nseq, slen, H, spec.scale, rope.cos, rope.sin)  # (rotated back in the stores)
        else:  # (saved qkv is unrotated: the forward ran with save=True, never as a prefill)
            ops.tokattn_bwd(qkv, do, dqkv, nseq, slen, H, spec.scale, rope.cos, rope.sin)
        dh1 = do
        ops.gemm_nt(dqkv, lw.wqkv, dh1, tb=True)        # d h1 = dqkv @ wqkv
        linear_wgrad(dqkv, h1, lg.wqkv, accumulate)
        del dqkv
        ops.rmsnorm_bwd(x, lw.n1, rstd1, dh1, dx2, dx, lg.n1, accumulate)
        saved[li] = None
        if on_layer_done is not None:
            on_layer_done(li)
    return dx


# --------------------------------------------------------------------------------------------------
# KV-cached decode (one new position per sequence)
# --------------------------------------------------------------------------------------------------
class KVState:
    """Preallocated per-layer K/V buffers [B,H,Lmax,hd] replacing DynamicCache's torch.cat growth
    (TF:cache_utils.py:127-147).  Attached to whatever cache object the caller passes (app.py hands us
    HF DynamicCache instances it created itself, app.py:56,64)."""

    def __init__(self, spec: StackSpec, B: int, capacity: int, like: torch.Tensor):
        self.spec, self.B, self.cap, self.len = spec, B, capacity, 0
        shape = (spec.L, B, spec.H, capacity, spec.hd)
        self.k = torch.empty(shape, dtype=like.dtype, device=like.device)
        self.v = torch.empty(shape, dtype=like.dtype, device=like.device)

    def reserve(self, need: int):
        if need <= self.cap:
            return
        cap = max(need, 2 * self.cap)
        for nm in ("k", "v"):
            old = getattr(self, nm)
            new = torch.empty((self.spec.L, self.B, self.spec.H, cap, self.spec.hd), dtype=old.dtype, device=old.device)
            new[:, :, :, : self.len].copy_(old[:, :, :, : self.len])
            setattr(self, nm, new)
        self.cap = cap


def stack_prefill(spec: StackSpec, W: StackTensors, x: torch.Tensor, nseq: int, slen: int, rope: RopeTable, kv: KVState,
                  folded=None):
    """Causal forward over a whole prompt from an EMPTY cache, storing K/V rows [0, slen)."""
    assert kv.len == 0
    kv.reserve(slen)
    qkvs: list = []
    y, _ = stack_forward(spec, W, x, nseq, slen, rope, save=False, kv_out=qkvs, folded=folded)
    for li, qkv in enumerate(qkvs):
        ops.kv_store_prefill(qkv, kv.k[li], kv.v[li], nseq, slen, spec.H, spec.hd, kv.cap)
    kv.len = slen
    return y


def stack_extend(spec: StackSpec, W: StackTensors, x: torch.Tensor, nseq: int, slen: int, rope: RopeTable, kv: KVState):
    """A chunk of ``slen`` > 1 new positions per sequence behind ``kv.len`` cached ones (a cache-carrying forward with q_len > 1:
    chunked prefill, TF:models/llama/modeling_llama.py:386-389 position offset + TF:integrations/sdpa_attention.py:79-166).
    Per layer: the chunk's q|k|v projection rotated at positions [n, n + slen), K/V appended to the cache, the cached K/V rows
    gathered in front of the chunk's rows, and the flash forward over the n + slen rows computing the chunk's query tiles only --
    the same ~9 launches per layer whatever ``slen`` is (the event-by-event form this replaces took 5 x slen).  Event-level
    stack (heads of 64)."""
    _check_heads(spec)
    assert spec.kind == "event"
    M, D = x.shape
    assert M == nseq * slen
    n, H, I, hd = kv.len, spec.H, spec.I, spec.hd
    stot = n + slen
    kv.reserve(stot)
    rope.ensure(stot)
    for li, lw in enumerate(W.layers):
        h1 = _empty((M, D), x)
        ops.rmsnorm_fwd(x, lw.n1, h1, None, spec.eps)
        qkv = _empty((M, 3 * D), x)
        if ops.rope_fused_ok(h1, hd):
            ops.gemm_rope(h1, lw.wqkv, qkv, rope.fused(), slen, n, hd)
        else:
            ops.gemm_nt(h1, lw.wqkv, qkv)
            ops.rope_(qkv, rope.cos, rope.sin, slen, n, H, hd, +1)
        ops.kv_store_rows(qkv, kv.k[li], kv.v[li], nseq, slen, H, hd, kv.cap, n)
        full = _empty((nseq * stot, 3 * D), x)
        ops.kv_gather_rows(kv.k[li], kv.v[li], full, nseq, n, stot, H, hd, kv.cap)
        full.view(nseq, stot, 3 * D)[:, n:].copy_(qkv.view(nseq, slen, 3 * D))
        o_full = _empty((nseq * stot, D), x)
        lse = _empty((nseq * H * ops.round_up(stot, 64),), x, torch.float32)
        ops.attn_fwd_tail(full, o_full, lse, nseq, stot, H, spec.scale, n)
        o = o_full.view(nseq, stot, D)[:, n:].reshape(M, D)
        x2 = _empty((M, D), x)
        ops.gemm_nt(o, lw.wo, x2, beta=1.0, res=x)
        h2 = _empty((M, D), x)
        ops.rmsnorm_fwd(x2, lw.n2, h2, None, spec.eps)
        a = _empty((M, I), x)
        if ops.swiglu_fused_ok(h2, I):
            ops.gemm_swiglu(h2, lw.wgu, None, a)
        else:
            gu = _empty((M, 2 * I), x)
            ops.gemm_nt(h2, lw.wgu, gu)
            ops.swiglu_fwd(gu, a)
        x3 = _empty((M, D), x)
        ops.gemm_nt(a, lw.wd, x3, beta=1.0, res=x2)
        x = x3
    kv.len = stot
    y = _empty((M, D), x)
    ops.rmsnorm_fwd(x, W.norm, y, None, spec.eps)
    return y


def fold_norm_weights(W: StackTensors, out=None):
    """[(wqkv * n1, wgu * n2) per layer]: the RMSNorm weights folded into the projections that follow them, for the
    decode path's one-launch norm + projection (mh_gemm_skinny with norm_eps).  Derived data: with ``out`` (a list this
    function returned earlier) the copies are rewritten in place, which is how a decode session follows weight updates."""
    if out is None:
        out = _FoldList((torch.empty_like(lw.wqkv), torch.empty_like(lw.wgu)) for lw in W.layers)
    if W.layers and W.layers[0].wqkv.is_cuda and all(lw.wqkv.is_contiguous() and lw.wgu.is_contiguous() for lw in W.layers):
        # ONE launch for the whole stack (the training step re-derives the fold after every optimizer step): the job table is
        # built once per list -- the matrices are views of the flat parameter buffer and the copies are rewritten in place
        jobs = getattr(out, "jobs", None)
        key = (W.layers[0].wqkv.data_ptr(), out[0][0].data_ptr())
        if jobs is None or jobs[0] != key:
            trip = []
            for lw, (fq, fg) in zip(W.layers, out):
                trip += [(lw.wqkv, lw.n1, fq), (lw.wgu, lw.n2, fg)]
            jobs = (key, ops.scale_cols_jobs(trip))
            if isinstance(out, _FoldList):
                out.jobs = jobs
        ops.scale_cols_batched(jobs[1], W.layers[0].wqkv.shape[1], W.layers[0].wqkv)
        return out
    for lw, (fq, fg) in zip(W.layers, out):
        fq.copy_(lw.wqkv.float() * lw.n1.float()[None, :])
        fg.copy_(lw.wgu.float() * lw.n2.float()[None, :])
    return out


class _FoldList(list):
    """fold_norm_weights' result: a list of (wqkv * n1, wgu * n2) that can carry the job table of its one-launch refresh"""
    jobs = None


def stack_decode(spec: StackSpec, W: StackTensors, x: torch.Tensor, rope: RopeTable, kv: KVState, pos_dev=None,
                 folded=None, final_norm: bool = True, x_ids=None, out: Optional[torch.Tensor] = None):
    """x [B, D]: one new position per sequence at index kv.len (q_len == 1 => no causal mask,
    TF:integrations/sdpa_attention.py:120).

    With ``pos_dev`` (device int32[1]) the kernels take the position from device memory instead -- the form a captured
    hipGraph replays (decode.py); capacity and rope table must already cover it and kv.len is left to the caller.
    bf16 with at most 64 sequences runs the projections on mh_gemm_skinny: 7 launches per layer (K/V append fused
    into the attention, gate|up and SwiGLU fused), 5 with ``folded`` (fold_norm_weights: the two RMSNorms ride on the
    q|k|v and gate|up projections); otherwise the general GEMM is used (9 launches + split-K reductions).
    ``final_norm=False`` returns the residual stream before the stack's last RMSNorm (decode.py folds it into lm_head).
    ``x_ids`` (folded form only): ``x`` is an embedding table and row b of the input is x[x_ids[b]] -- the lookup happens
    inside the first layer's projections."""
    _check_heads(spec)
    B, D = (x_ids.shape[0], x.shape[1]) if x_ids is not None else x.shape
    assert x_ids is None or folded is not None
    H, I, hd = spec.H, spec.I, spec.hd
    pos = kv.len if pos_dev is None else 0
    if pos_dev is None:
        kv.reserve(pos + 1)
        rope.ensure(pos + 1)
    fused = (x_ids is not None) or (ops.skinny_ok(x, D) and ops.skinny_ok(x, I))
    for li, lw in enumerate(W.layers):
        if fused and folded is not None:
            wqkv_n, wgu_n = folded[li]
            ids = x_ids if li == 0 else None  # layer 0 may read its input rows straight from the embedding table
            qkv = _empty((B, 3 * D), x)
            ops.gemm_skinny(x, wqkv_n, qkv, norm_eps=spec.eps, row_ids=ids)
            o = _empty((B, D), x)
            ops.attn_decode_append(qkv, rope.cos, rope.sin, kv.k[li], kv.v[li], o, B, H, hd, kv.cap, pos, spec.scale, pos_dev)
            x2 = _empty((B, D), x)
            ops.gemm_skinny(o, lw.wo, x2, res=x, res_ids=ids)
            a = _empty((B, I), x)
            ops.gemm_skinny(x2, wgu_n, a, mode=ops.SKINNY_GATEUP, norm_eps=spec.eps)
            x3 = _empty((B, D), x)
            ops.gemm_skinny(a, lw.wd, x3, res=x2)
            x = x3
            continue
        if fused:
            h1 = _empty((B, D), x)
            ops.rmsnorm_fwd(x, lw.n1, h1, None, spec.eps)
            qkv = _empty((B, 3 * D), x)
            ops.gemm_skinny(h1, lw.wqkv, qkv)
            o = h1
            ops.attn_decode_append(qkv, rope.cos, rope.sin, kv.k[li], kv.v[li], o, B, H, hd, kv.cap, pos, spec.scale, pos_dev)
            x2 = _empty((B, D), x)
            ops.gemm_skinny(o, lw.wo, x2, res=x)
            h2 = o
            ops.rmsnorm_fwd(x2, lw.n2, h2, None, spec.eps)
            a = _empty((B, I), x)
            ops.gemm_skinny(h2, lw.wgu, a, mode=ops.SKINNY_GATEUP)  # gate|up never materialised
            x3 = _empty((B, D), x)
            ops.gemm_skinny(a, lw.wd, x3, res=x2)
            x = x3
            continue
        h1 = _empty((B, D), x)
        ops.rmsnorm_fwd(x, lw.n1, h1, None, spec.eps)
        qkv = _empty((B, 3 * D), x)
        ops.gemm_nt(h1, lw.wqkv, qkv)
        ops.kv_append(qkv, rope.cos, rope.sin, kv.k[li], kv.v[li], B, H, hd, kv.cap, pos, pos_dev)
        o = h1
        ops.attn_decode(qkv, kv.k[li], kv.v[li], o, B, H, hd, kv.cap, pos + 1, spec.scale, pos_dev)
        x2 = _empty((B, D), x)
        ops.gemm_nt(o, lw.wo, x2, beta=1.0, res=x)
        h2 = o
        ops.rmsnorm_fwd(x2, lw.n2, h2, None, spec.eps)
        gu = _empty((B, 2 * I), x)
        ops.gemm_nt(h2, lw.wgu, gu)
        a = _empty((B, I), x)
        ops.swiglu_fwd(gu, a)
        x3 = _empty((B, D), x)
        ops.gemm_nt(a, lw.wd, x3, beta=1.0, res=x2)
        x = x3
    if pos_dev is None:
        kv.len = pos + 1
    if not final_norm:
        return x
    y = out if out is not None else _empty((B, D), x)  # (`out`: the caller's buffer -- a decode session's `hidden` -- no copy)
    ops.rmsnorm_fwd(x, W.norm, y, None, spec.eps)
    return y
