"""Tensor-level wrappers over the C-ABI (include/midihip.h).  torch is only the memory/stream plumbing
here: every function passes raw device pointers of caller-owned tensors to libmidihip.so on the current
HIP stream.  Nothing in this module computes on the host and nothing falls back to torch math.
"""
from __future__ import annotations

from typing import Optional

import os
import threading

import torch

from .lib import MH_BF16, MH_F32, lib

_DT = {torch.float32: MH_F32, torch.bfloat16: MH_BF16}


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}: the HIP path computes in float32 or bfloat16") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise RuntimeError("HIP kernels need device tensors (there is no CPU implementation of this path)")
    return t.data_ptr()


def _rowmajor(t: torch.Tensor) -> int:
    """leading dimension (elements) of a 2-D row-major view"""
    assert t.dim() == 2 and t.stride(1) == 1, f"need a row-major 2-D view, got strides {t.stride()}"
    return t.stride(0)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ---------------------------------------------------------------------------------------------- GEMM
class _OptionCache(threading.local):
    """per-thread mirrors of mh_get_option values (the C-ABI's options are thread-local, so are their caches)"""
    gemm_variant = None
    attn_v3 = None  # which forms of the event-level attention kernels run (attn_bwd)


_tl = _OptionCache()




class ab_library:
    """``with ops.ab_library():`` (tests, tools): run on libmidihip_ab.so, the build that also holds the first-form attention
    kernels, the 128x128 bf16 GEMM and the ablation builds; the cached option values follow the library in use"""

    def __enter__(self):
        from .lib import use_ab
        self._ctx = use_ab()
        self._ctx.__enter__()
        self._saved = (_tl.gemm_variant, _tl.attn_v3)
        _tl.gemm_variant = _tl.attn_v3 = None
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        _tl.gemm_variant, _tl.attn_v3 = self._saved
        return False


def set_option(name: str, value: int) -> None:
    """runtime kernel selection (see mh_set_option in include/midihip.h)"""
    lib().call("mh_set_option", name.encode(), int(value))
    if name == "gemm":
        _tl.gemm_variant = int(value)
    if name == "attn_v3":
        _tl.attn_v3 = int(value)


def get_option(name: str) -> int:
    if name == "gemm":
        if _tl.gemm_variant is None:
            _tl.gemm_variant = lib().cdll.mh_get_option(b"gemm")
        return _tl.gemm_variant
    return lib().cdll.mh_get_option(name.encode())


def _pick_splitk(M: int, N: int, K: int) -> int:
    """Split the contraction when the output has too few tiles to fill 256 CUs (the weight-gradient shapes):
    aim at >= 512 workgroups, keep >= 512 contraction elements per slice."""
    if _tl.gemm_variant is None:
        _tl.gemm_variant = lib().cdll.mh_get_option(b"gemm")
    bm = bn = 256 if _tl.gemm_variant != 0 else 128
    tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    per_cu = 1 if _tl.gemm_variant != 0 else 2   # resident workgroups per CU of the active kernel
    if tiles >= 192 * per_cu or K < 1024:
        return 1
    # fill the 256 CUs once (or twice for the two-per-CU kernel) but never spill a few workgroups into an extra
    # round: 48 tiles x 6 slices = 288 workgroups ran at 590 TFLOP/s where 48 x 5 = 240 fits one round
    s = int(max(1, min(64, (256 * per_cu) // tiles, K // 512)))
    while True:  # slices are whole 32-deep K-steps: shrink the count until none is left empty
        kps = -(-(-(-K // s)) // 32) * 32
        s2 = -(-K // kps)
        if s2 == s:
            return s
        s = s2


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, K: Optional[int] = None, alpha: float = 1.0,
            beta: float = 0.0, res: Optional[torch.Tensor] = None, splitk: int = 0, ta: bool = False,
            tb: bool = False) -> torch.Tensor:
    """out[M,N] = alpha * A @ B^T + beta * res   (res defaults to `out` when beta != 0).

    A is a[M,K] (or, with ``ta``, stored contraction-major as a[K,M]); B is b[N,K] (with ``tb``: b[K,N]).
    The contraction-major forms are what dgrad / wgrad present; bf16 feeds them to the kernel as they lie
    (LDS transpose reads), the fp32 verification mode re-lays them out with mh_transpose first."""
    M, N = out.shape
    if (ta or tb) and out.dtype != torch.bfloat16:
        K = K if K is not None else (a.shape[0] if ta else a.shape[1])
        if ta:
            a = transpose(a[:K])  # [M, K rounded up to 8], zero padded
        if tb:
            b = transpose(b[:K])
        ta = tb = False
    if K is None:
        K = a.shape[0] if ta else a.shape[1]
    ka, ma = (a.shape[0], a.shape[1]) if ta else (a.shape[1], a.shape[0])
    kb, nb = (b.shape[0], b.shape[1]) if tb else (b.shape[1], b.shape[0])
    assert ma >= M and nb >= N and ka >= K and kb >= K, (a.shape, b.shape, out.shape, K, ta, tb)
    assert a.dtype == b.dtype == out.dtype
    if beta != 0.0 and res is None:
        res = out
    if splitk <= 0:
        splitk = _pick_splitk(M, N, K)
    ws = None
    if splitk > 1:
        ws = torch.empty((splitk, M, N), dtype=torch.float32, device=out.device)
    prof = gemm_profile
    if prof is not None:  # bench.py: HIP events on the launch stream around every projection GEMM
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib().call("mh_gemm", _p(a), _rowmajor(a), int(ta), _p(b), _rowmajor(b), int(tb), _p(out), _rowmajor(out), _p(res),
               _rowmajor(res) if res is not None else 0, M, N, K, alpha, beta, dt(out), splitk, _p(ws), _stream())
    if splitk > 1:
        lib().call("mh_gemm_splitk_reduce", _p(ws), _p(out), _rowmajor(out), _p(res),
                   _rowmajor(res) if res is not None else 0, M, N, splitk, alpha, beta, dt(out), _stream())
    if prof is not None:  # (the window includes the split-K reduction: it is part of what the projection costs)
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, (M, N, K, splitk, int(ta), int(tb))))
    return out


_FUSE = int(os.environ.get("MH_FUSE_EPILOGUES", "7"))  # GEMM epilogues: bit 0 SwiGLU forward, bit 1 SwiGLU backward, bit 2 RoPE (A/B runs)


def rope_fused_ok(x: torch.Tensor, hd: int) -> bool:
    """whether mh_gemm_rope serves the q|k|v projection + RoPE of the event-level stack (else mh_gemm, then mh_rope)"""
    return x.dtype == torch.bfloat16 and hd == 64 and get_option("gemm") != 0 and (_FUSE & 4) != 0


def norm_fold_ok(x: torch.Tensor, D: int, hd: int, I: int) -> bool:
    """whether the forward-only block can fold its two RMSNorms around the projections (mh_gemm_rowss producers, mh_row_rstd,
    mh_gemm_rope_scaled / mh_gemm_swiglu_scaled consumers): bf16, event-level heads of 64, the fused epilogues available,
    whole 64-column chunks and 4-row groups"""
    return (rope_fused_ok(x, hd) and swiglu_fused_ok(x, I) and D % 64 == 0 and x.shape[0] % 4 == 0 and get_option("gemm_k64") != 0
            and os.environ.get("MH_NORM_FOLD", "1") != "0")


def gemm_rowss(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, rowss: torch.Tensor, res: Optional[torch.Tensor] = None):
    """out = a @ b^T (+ res) and rowss[N // 64, M] (fp32) = per-64-column sums of squares of the stored rows of `out`"""
    M, N = out.shape
    K = a.shape[1]
    assert b.shape == (N, K) and a.shape[0] == M and N % 64 == 0 and rowss.shape == (N // 64, M) and rowss.dtype == torch.float32
    assert rowss.is_contiguous() and (res is None or res.shape == out.shape)
    prof = gemm_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib().call("mh_gemm_rowss", _p(a), _rowmajor(a), _p(b), _rowmajor(b), _p(out), _rowmajor(out), _p(res),
               _rowmajor(res) if res is not None else 0, _p(rowss), M, N, K, dt(out), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, (M, N, K, 1, 0, 0, "+rowss")))
    return out


def row_rstd(rstd: torch.Tensor, D: int, eps: float, x: Optional[torch.Tensor] = None, parts: Optional[torch.Tensor] = None):
    """rstd[M] (fp32) = rsqrt(mean(row^2) + eps) from the rows ``x`` [M, D] or from ``parts`` = gemm_rowss's [D // 64, M] sums"""
    M = rstd.shape[0]
    assert (x is None) != (parts is None) and rstd.dtype == torch.float32 and rstd.is_contiguous()
    if parts is not None:
        assert parts.shape[1] == M and parts.dtype == torch.float32 and parts.is_contiguous()
        lib().call("mh_row_rstd", None, 0, _p(parts), parts.shape[0], M, D, eps, _p(rstd), MH_BF16, _stream())
    else:
        assert x.shape == (M, D)
        lib().call("mh_row_rstd", _p(x), _rowmajor(x), None, 0, M, D, eps, _p(rstd), dt(x), _stream())
    return rstd


def gemm_rope(x: torch.Tensor, wqkv: torch.Tensor, qkv: torch.Tensor, table: torch.Tensor, S: int, pos0: int, hd: int,
              rowscale: Optional[torch.Tensor] = None):
    """qkv = x @ wqkv^T with q and k rotated in the projection's epilogue (table: RopeTable.fused()); ``rowscale`` (fp32 [M]):
    every row of the product times rowscale[m] first -- RMSNorm with its weight folded into wqkv by the caller"""
    M, K = x.shape
    N = wqkv.shape[0]
    assert qkv.shape == (M, N) and wqkv.shape[1] == K and table.dtype == torch.bfloat16 and table.shape[1] == 96
    prof = gemm_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if rowscale is not None:
        assert rowscale.shape == (M,) and rowscale.dtype == torch.float32 and rowscale.is_contiguous()
        lib().call("mh_gemm_rope_scaled", _p(x), _rowmajor(x), _p(wqkv), _rowmajor(wqkv), _p(qkv), _rowmajor(qkv), _p(table),
                   table.shape[0], S, pos0, hd, _p(rowscale), M, N, K, dt(x), _stream())
    else:
        lib().call("mh_gemm_rope", _p(x), _rowmajor(x), _p(wqkv), _rowmajor(wqkv), _p(qkv), _rowmajor(qkv), _p(table),
                   table.shape[0], S, pos0, hd, M, N, K, dt(x), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, (M, N, K, 1, 0, 0, "+rope")))
    return qkv


def swiglu_fused_ok(x: torch.Tensor, I: int) -> bool:
    """whether mh_gemm_swiglu serves the gate|up projection + SwiGLU (else mh_gemm, then mh_swiglu_fwd)"""
    return x.dtype == torch.bfloat16 and I % 128 == 0 and get_option("gemm") != 0 and (_FUSE & 1) != 0


def gemm_swiglu(x: torch.Tensor, wgu: torch.Tensor, gu: Optional[torch.Tensor], a: torch.Tensor,
                rowscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gu [M, 2I] = x @ wgu^T (wgu = [gate; up], [2I, K]) and a [M, I] = silu(gate) * up; gu=None: forward only, gate|up is
    not written (nothing will backpropagate); ``rowscale`` (fp32 [M]): rows of the product scaled first (folded RMSNorm)"""
    M, K = x.shape
    I = a.shape[1]
    assert wgu.shape == (2 * I, K) and a.shape[0] == M and x.dtype == wgu.dtype == a.dtype
    assert gu is None or (gu.shape == (M, 2 * I) and gu.dtype == x.dtype)
    prof = gemm_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if rowscale is not None:
        assert rowscale.shape == (M,) and rowscale.dtype == torch.float32 and rowscale.is_contiguous()
        lib().call("mh_gemm_swiglu_scaled", _p(x), _rowmajor(x), _p(wgu), _rowmajor(wgu), _p(gu), _rowmajor(gu) if gu is not None else 0,
                   _p(a), _rowmajor(a), _p(rowscale), M, I, K, dt(x), _stream())
    else:
        lib().call("mh_gemm_swiglu", _p(x), _rowmajor(x), _p(wgu), _rowmajor(wgu), _p(gu), _rowmajor(gu) if gu is not None else 0,
                   _p(a), _rowmajor(a), M, I, K, dt(x), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * 2 * I * K, (M, 2 * I, K, 1, 0, 0, "+swiglu")))
    return a


def dswiglu_ok(dx: torch.Tensor, I: int) -> bool:
    """whether mh_gemm_dswiglu serves down_proj's dgrad + SwiGLU backward (else mh_gemm, then mh_swiglu_bwd)"""
    return dx.dtype == torch.bfloat16 and I % 8 == 0 and get_option("gemm") != 0 and (_FUSE & 2) != 0


def gemm_dswiglu(dx: torch.Tensor, wd: torch.Tensor, gu: torch.Tensor, dgu: torch.Tensor,
                 rowscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dgu = SwiGLU'(gu) applied to (dx @ wd): dx [M, D], wd [D, I] (the down_proj weight, read contraction-major),
    gu / dgu [M, 2I]; ``rowscale`` (fp32 [M]): row m of the result times rowscale[m] (the folded RMSNorm's d z)"""
    M, K = dx.shape
    I = wd.shape[1]
    assert wd.shape[0] == K and gu.shape == (M, 2 * I) and dgu.shape == (M, 2 * I) and dx.dtype == wd.dtype == gu.dtype
    prof = gemm_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if rowscale is not None:
        assert rowscale.shape == (M,) and rowscale.dtype == torch.float32 and rowscale.is_contiguous()
        lib().call("mh_gemm_dswiglu_scaled", _p(dx), _rowmajor(dx), _p(wd), _rowmajor(wd), _p(gu), _rowmajor(gu), _p(dgu),
                   _rowmajor(dgu), _p(rowscale), M, I, K, dt(dx), _stream())
    else:
        lib().call("mh_gemm_dswiglu", _p(dx), _rowmajor(dx), _p(wd), _rowmajor(wd), _p(gu), _rowmajor(gu), _p(dgu), _rowmajor(dgu),
                   M, I, K, dt(dx), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * I * K, (M, I, K, 1, 0, 1, "+dswiglu")))
    return dgu


SKINNY_PLAIN, SKINNY_GATEUP = 0, 1


def gemm_skinny(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, mode: int = SKINNY_PLAIN,
                res: Optional[torch.Tensor] = None, norm_eps: float = 0.0, row_ids: Optional[torch.Tensor] = None,
                res_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """decode-step projection, M <= 64 rows, bf16 (mh_gemm_skinny): out[M,N] = a @ w[N,K]^T (+ res), or with
    SKINNY_GATEUP w = [gate; up] ([2N,K]) and out[M,N] = silu(a @ gate^T) * (a @ up^T).  norm_eps > 0 scales row m of
    the product by rsqrt(mean(a[m]^2) + norm_eps) first (RMSNorm with its weight folded into w).  row_ids / res_ids
    (int64 [M]): `a` / `res` are tables and row m is their row ids[m] (embedding lookup folded in)."""
    M, N = out.shape
    K = w.shape[1]
    assert a.shape[1] == K and (row_ids is not None or a.shape[0] == M), (a.shape, out.shape)
    assert w.shape[0] == (2 * N if mode == SKINNY_GATEUP else N), (a.shape, w.shape, out.shape, mode)
    for t in (row_ids, res_ids):
        assert t is None or (t.dtype == torch.int64 and t.is_contiguous() and t.numel() == M)
    # (the kernel addresses with 32-bit element offsets: the tables behind row_ids / res_ids must stay below 2^31 elements)
    assert a.shape[0] * _rowmajor(a) < 2 ** 31 and (res is None or res.shape[0] * _rowmajor(res) < 2 ** 31), "gemm_skinny: table too large"
    lib().call("mh_gemm_skinny", _p(a), _rowmajor(a), _p(w), _rowmajor(w), _p(out), _rowmajor(out), _p(res),
               _rowmajor(res) if res is not None else 0, mode, norm_eps, _p(row_ids), _p(res_ids), M, N, K, dt(out),
               _stream())
    return out


def skinny_ok(x: torch.Tensor, K: int) -> bool:
    """whether mh_gemm_skinny serves this decode projection (else mh_gemm)"""
    return x.dtype == torch.bfloat16 and x.shape[0] <= 64 and K % 256 == 0


gemm_profile = None  # set to a list to collect (start_event, end_event, flops, shape) per GEMM launch


def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None, pad_to: int = 8) -> torch.Tensor:
    """x[R,C] -> out[C, ld>=R]; a fresh `out` has ld = R rounded up to `pad_to` with zeroed padding."""
    R, C = x.shape
    if out is None:
        Rp = round_up(R, pad_to)
        out = torch.empty((C, Rp), dtype=x.dtype, device=x.device)
        if Rp != R:
            out[:, R:].zero_()
    lib().call("mh_transpose", _p(x), _rowmajor(x), _p(out), _rowmajor(out), R, C, dt(x), _stream())
    return out


def collate_windows(tokens_i16: torch.Tensor, win_start: torch.Tensor, win_len: torch.Tensor, out: torch.Tensor, pad_id: int):
    """out[B, L, T] int64 <- windows of the device-resident int16 corpus tokens_i16[N, T], padded with pad_id"""
    B, L, T = out.shape
    assert tokens_i16.dtype == torch.int16 and tokens_i16.is_contiguous() and tokens_i16.shape[1] == T
    assert win_start.dtype == torch.int64 and win_len.dtype == torch.int64 and out.dtype == torch.int64 and out.is_contiguous()
    lib().call("mh_collate_windows", _p(tokens_i16), tokens_i16.shape[0], _p(win_start), _p(win_len), _p(out), B, L, T, pad_id,
               _stream())
    return out


def augment_piece_stats(tokens_i16: torch.Tensor, piece_off: torch.Tensor, tab: torch.Tensor, stats: torch.Tensor):
    """stats[P, 130] int32 <- the whole-file facts MIDITokenizer.augment's two file-level rules need (mh_augment_piece_stats)"""
    P = piece_off.numel() - 1
    assert tokens_i16.dtype == torch.int16 and tokens_i16.is_contiguous() and piece_off.dtype == torch.int64
    assert tab.dtype == torch.int32 and stats.dtype == torch.int32 and stats.is_contiguous() and stats.shape == (P, 130)
    lib().call("mh_augment_piece_stats", _p(tokens_i16), _p(piece_off), P, _p(tab), _p(stats), _stream())
    return stats


def augment_collate_windows(tokens_i16: torch.Tensor, win_start: torch.Tensor, win_len: torch.Tensor, win_piece: torch.Tensor,
                            shifts: torch.Tensor, stats: torch.Tensor, tab: torch.Tensor, out: torch.Tensor, pad_id: int):
    """collate_windows with every window augmented as MIDITokenizer.augment would have augmented its file (shifts[B, 6] int32:
    pitch, velocity, cc value, bpm, track, channel)"""
    B, L, T = out.shape
    assert tokens_i16.dtype == torch.int16 and tokens_i16.is_contiguous() and tokens_i16.shape[1] == T
    assert win_start.dtype == torch.int64 and win_len.dtype == torch.int64 and win_piece.dtype == torch.int64
    assert shifts.dtype == torch.int32 and shifts.shape == (B, 6) and shifts.is_contiguous() and stats.dtype == torch.int32
    assert out.dtype == torch.int64 and out.is_contiguous() and tab.dtype == torch.int32
    lib().call("mh_augment_collate_windows", _p(tokens_i16), tokens_i16.shape[0], _p(win_start), _p(win_len), _p(win_piece),
               _p(shifts), _p(stats), _p(tab), _p(out), B, L, T, pad_id, _stream())
    return out


# ---------------------------------------------------------------------------------------- embeddings
def embed_sum_fwd(tok: torch.Tensor, table: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    M, T = tok.shape
    assert tok.dtype == torch.int64 and tok.is_contiguous() and table.is_contiguous()
    lib().call("mh_embed_sum_fwd", _p(tok), _p(table), _p(out), M, T, table.shape[0], table.shape[1], dt(table), _stream())
    return out


def concat_tok_fwd(hidden: torch.Tensor, tok: torch.Tensor, table: torch.Tensor, out: torch.Tensor, T: int) -> torch.Tensor:
    """out[M,T,D]: row 0 = hidden, rows 1..T-1 = table[tok[:, :T-1]]; tok may be a strided [M, >=T-1] view."""
    M = hidden.shape[0]
    assert tok.dtype == torch.int64 and tok.stride(1) == 1
    lib().call("mh_concat_tok_fwd", _p(hidden), _p(tok), tok.stride(0), _p(table), _p(out), M, T, table.shape[0],
               table.shape[1], dt(table), _stream())
    return out


def embed_scatter_bwd(tok: torch.Tensor, T: int, dout: torch.Tensor, rows_per_m: int, jstride: int, j0: int,
                      dtable_f32: torch.Tensor, pad_id: int) -> None:
    M = tok.shape[0]
    assert tok.dtype == torch.int64 and tok.stride(1) == 1 and dtable_f32.dtype == torch.float32
    V, D = dtable_f32.shape
    lib().call("mh_embed_scatter_bwd", _p(tok), tok.stride(0), T, _p(dout), rows_per_m, jstride, j0, _p(dtable_f32), M, V, D,
               pad_id, dt(dout), _stream())


def token_segments(tok: torch.Tensor, V: int, row_mul: Optional[int] = None, col_mul: int = 1, add: int = 0):
    """Index preparation for embed_segment_bwd (mh_token_segments: a counting sort on the device, three launches): tok is the id
    matrix [n_rows, n_cols] (int64, unit column stride, any row stride -- a column slice of a wider matrix is fine) or a flat
    vector.  -> (src_rows, seg_start): seg_start[v] (V + 1 entries) = occurrences with id < v, src_rows[p] = r * row_mul +
    j * col_mul + add for the occurrence (row r, column j) placed at p.  The default mapping (row_mul = n_cols, 1, 0) makes
    src_rows the occurrence indices grouped by id (`order`)."""
    if tok.dim() == 1:
        tok = tok.view(-1, 1)
    assert tok.dim() == 2 and tok.dtype == torch.int64 and tok.stride(1) == 1
    n_rows, n_cols = tok.shape
    if row_mul is None:
        row_mul = n_cols
    n = n_rows * n_cols
    chunk = lib().cdll.mh_token_segments_chunk()
    work = torch.empty((((n + chunk - 1) // chunk) + 1) * (V + 1), dtype=torch.int32, device=tok.device)
    src = torch.empty(n, dtype=torch.int64, device=tok.device)
    seg = torch.empty(V + 1, dtype=torch.int64, device=tok.device)
    lib().call("mh_token_segments", _p(tok), tok.stride(0), n_rows, n_cols, V, row_mul, col_mul, add, _p(src), _p(seg), _p(work),
               _stream())
    return src, seg


def embed_segment_bwd(src_rows: torch.Tensor, seg_start: torch.Tensor, dout: torch.Tensor, ld: int,
                      dtable_f32: torch.Tensor, pad_id: int) -> None:
    V, D = dtable_f32.shape
    assert src_rows.dtype == torch.int64 and seg_start.dtype == torch.int64 and seg_start.numel() == V + 1
    assert src_rows.is_contiguous() and dtable_f32.dtype == torch.float32
    lib().call("mh_embed_segment_bwd", _p(src_rows), _p(seg_start), _p(dout), ld, _p(dtable_f32), V, D, src_rows.numel(), pad_id,
               dt(dout), _stream())


def cast_from_f32(src: torch.Tensor, dst: torch.Tensor, accumulate: bool) -> None:
    assert src.dtype == torch.float32 and src.numel() == dst.numel() and dst.is_contiguous()
    lib().call("mh_cast_from_f32", _p(src), _p(dst), src.numel(), int(accumulate), dt(dst), _stream())


def copy_rows(src: torch.Tensor, src_ld: int, dst: torch.Tensor, dst_ld: int, M: int, D: int, accumulate: bool = False) -> None:
    lib().call("mh_copy_rows", _p(src), src_ld, _p(dst), dst_ld, M, D, int(accumulate), dt(dst), _stream())


# ------------------------------------------------------------------------------------------- RMSNorm
def rmsnorm_fwd(x, w, y, rstd, eps: float):
    M, D = x.shape
    lib().call("mh_rmsnorm_fwd", _p(x), _p(w), _p(y), _p(rstd), M, D, eps, dt(x), _stream())
    return y


def rmsnorm_bwd(x, w, rstd, dy, dres, dx, dw: torch.Tensor, accumulate: bool):
    """dx = d(norm)(dy) + dres; dw (+)= column sums."""
    M, D = x.shape
    nblk = lib().cdll.mh_rmsnorm_bwd_blocks(M)
    partial = torch.empty((nblk, D), dtype=torch.float32, device=x.device)
    lib().call("mh_rmsnorm_bwd", _p(x), _p(w), _p(rstd), _p(dy), _p(dres), _p(dx), _p(partial), M, D, dt(x), _stream())
    lib().call("mh_colsum", _p(partial), nblk, _p(dw), D, int(accumulate), dt(dw), _stream())
    return dx


def rmsnorm_bwd_folded(x, rstd, t, dres, dx):
    """the backward of a FOLDED RMSNorm: dx = t - x (rstd^2 / D) rowdot(t, x) + dres, t = d z @ W' (see mh_rmsnorm_bwd_folded)"""
    M, D = x.shape
    lib().call("mh_rmsnorm_bwd_folded", _p(x), _p(rstd), _p(t), _p(dres), _p(dx), M, D, dt(x), _stream())
    return dx


def scale_cols(W: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[n, k] = W[n, k] * w[k] (the norm weight folded into the projection that follows it)"""
    N, K = W.shape
    assert w.shape == (K,) and out.shape == (N, K) and W.dtype == w.dtype == out.dtype
    lib().call("mh_scale_cols", _p(W), _rowmajor(W), _p(w), _p(out), _rowmajor(out), N, K, dt(W), _stream())
    return out


def scale_cols_jobs(triples) -> torch.Tensor:
    """the device-side job table of scale_cols_batched for [(W, w, out), ...] (contiguous [rows, K] matrices of one K and dtype)"""
    rows = []
    for W, w, out in triples:
        assert W.is_contiguous() and out.is_contiguous() and W.shape == out.shape and w.shape == (W.shape[1],)
        rows.append([W.data_ptr(), w.data_ptr(), out.data_ptr(), W.shape[0]])
    return torch.tensor(rows, dtype=torch.int64, device=triples[0][0].device)


def scale_cols_batched(jobs: torch.Tensor, K: int, like: torch.Tensor) -> None:
    """out_j[n, k] = W_j[n, k] * w_j[k] for every job of the table, one launch"""
    lib().call("mh_scale_cols_batched", _p(jobs), jobs.shape[0], K, dt(like), _stream())


def gemm_nt_scaled(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, rowscale: torch.Tensor) -> torch.Tensor:
    """out[M, N] = rowscale[:, None] * (a @ b^T): the plain projection with the row scale of a folded RMSNorm"""
    M, N = out.shape
    K = a.shape[1]
    assert b.shape == (N, K) and a.shape[0] == M and rowscale.shape == (M,) and rowscale.dtype == torch.float32
    prof = gemm_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib().call("mh_gemm_nt_scaled", _p(a), _rowmajor(a), _p(b), _rowmajor(b), _p(out), _rowmajor(out), _p(rowscale), M, N, K,
               dt(a), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, (M, N, K, 1, 0, 0, "+rowscale")))
    return out


def wgrad_folded(dz: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, wnorm: torch.Tensor, W: torch.Tensor,
                 dnorm: torch.Tensor, accumulate: bool) -> None:
    """The weight gradient of a projection behind a FOLDED RMSNorm: G' = dz^T @ x (split-K over the rows, operands read as they
    lie), then in the reduction of the fp32 partials  dw (+)= G' * wnorm[None, :]  and  dnorm (+)= colsum(G' * W)
    (mh_gemm_splitk_reduce_fold + mh_colsum).  dz [M, N], x [M, K], dw / W [N, K], wnorm / dnorm [K]."""
    M = dz.shape[0]
    N, K = dw.shape
    assert dz.shape[1] == N and x.shape == (M, K) and W.shape == (N, K) and wnorm.shape == (K,) and dnorm.shape == (K,)
    splitk = max(2, _pick_splitk(N, K, M))   # (the fold's chain rule lives in the reduction: always at least two slices)
    ws = torch.empty((splitk, N, K), dtype=torch.float32, device=dw.device)
    prof = gemm_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib().call("mh_gemm", _p(dz), _rowmajor(dz), 1, _p(x), _rowmajor(x), 1, _p(dw), _rowmajor(dw), None, 0, N, K, M, 1.0, 0.0,
               dt(dw), splitk, _p(ws), _stream())
    nblk = lib().cdll.mh_splitk_fold_blocks(N)
    colpart = torch.empty((nblk, K), dtype=torch.float32, device=dw.device)
    lib().call("mh_gemm_splitk_reduce_fold", _p(ws), _p(dw), _rowmajor(dw), _p(dw) if accumulate else None, _rowmajor(dw) if accumulate else 0,
               N, K, splitk, 1.0, 1.0 if accumulate else 0.0, _p(wnorm), _p(W), _rowmajor(W), _p(colpart), dt(dw), _stream())
    lib().call("mh_colsum", _p(colpart), nblk, _p(dnorm), K, int(accumulate), dt(dnorm), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, (N, K, M, splitk, 1, 1, "+fold")))


# ---------------------------------------------------------------------------------------------- RoPE
def rope_(qkv: torch.Tensor, cos_t: torch.Tensor, sin_t: torch.Tensor, S: int, pos0: int, H: int, hd: int, direction: int = 1):
    M = qkv.shape[0]
    assert qkv.is_contiguous() and qkv.shape[1] == 3 * H * hd
    assert cos_t.shape[0] >= pos0 + min(S, M), "RoPE table too short"
    lib().call("mh_rope", _p(qkv), _p(cos_t), _p(sin_t), M, S, pos0, H, hd, direction, dt(qkv), _stream())
    return qkv


# ----------------------------------------------------------------------------------------- attention
def attn_fwd(qkv, o, lse, B: int, S: int, H: int, scale: float):
    """Event-level causal flash attention.  bf16: the third form of the MFMA kernels (attention_mfma3.hip) reads V^T out of
    the row-major V tile with transpose reads; only the first form (mh_set_option("attn_v3", 0), kept for A/B runs) needs
    the prepared [B,H,64,Sp] copy.  fp32: the plain verification kernel."""
    # (A/B runs: the environment variables MH_ATTN_V3 / MH_ATTN_V3_WPS set every host thread's initial value inside the
    #  library; set_option / mh_set_option act on the CALLING thread only -- autograd's backward runs on its own thread, so an
    #  A/B of a backward kernel through loss.backward() goes by the environment, or calls ops.attn_bwd directly)
    if _tl.attn_v3 is None:
        _tl.attn_v3 = get_option("attn_v3")
    vt = None
    if qkv.dtype == torch.bfloat16 and (_tl.attn_v3 & 17) != 17:
        Sp = round_up(S, 64)
        vt = torch.empty((B * H * 64 * Sp,), dtype=qkv.dtype, device=qkv.device)
        lib().call("mh_attn_prep_fwd", _p(qkv), _p(vt), B, S, H, dt(qkv), _stream())
    lib().call("mh_attn_fwd", _p(qkv), _p(vt), _p(o), _p(lse), B, S, H, scale, dt(qkv), _stream())
    return o


def attn_bwd_scaled_ok(qkv: torch.Tensor) -> bool:
    """whether attn_bwd serves ``rowscale`` (the one-call third form of the bf16 backward kernels)"""
    if _tl.attn_v3 is None:
        _tl.attn_v3 = get_option("attn_v3")
    return qkv.dtype == torch.bfloat16 and (_tl.attn_v3 & 46) == 46


def attn_bwd(qkv, o, dout, lse, dqkv, B: int, S: int, H: int, scale: float, cos_t=None, sin_t=None, rowscale=None):
    """cos_t/sin_t: return the gradient with respect to the UNROTATED q, k (see mh_attn_bwd); ``rowscale`` (fp32 [B * S]): row m
    of dqkv times rowscale[m] in the kernels' stores (the folded RMSNorm's d z; attn_bwd_scaled_ok)"""
    if _tl.attn_v3 is None:
        _tl.attn_v3 = get_option("attn_v3")
    Sp = round_up(S, 64)
    fused = qkv.dtype == torch.bfloat16 and (_tl.attn_v3 & 14) == 14 and (_tl.attn_v3 & 32) != 0
    delta = torch.empty(((2 if fused else 1) * B * H * Sp,), dtype=torch.float32, device=qkv.device)
    if rowscale is not None:
        assert fused and rowscale.shape == (B * S,) and rowscale.dtype == torch.float32 and rowscale.is_contiguous()
        lib().call("mh_attn_bwd_o_scaled", _p(qkv), _p(o), _p(dout), _p(lse), _p(delta), _p(dqkv), _p(rowscale), B, S, H, scale,
                   _p(cos_t), _p(sin_t), dt(qkv), _stream())
        return dqkv
    if fused:
        # (default) one call: the dQ kernel computes delta from its own rows and hands it (and -lse * log2 e) to the dK/dV kernel
        lib().call("mh_attn_bwd_o", _p(qkv), _p(o), _p(dout), _p(lse), _p(delta), _p(dqkv), B, S, H, scale, _p(cos_t), _p(sin_t),
                   dt(qkv), _stream())
        return dqkv
    qt = kt = dot = None
    # (third form with transpose reads, the default: dQ and dK/dV take Q^T, K^T, dO^T out of the row-major tiles in LDS --
    #  no [B,H,64,Sp] copies, mh_attn_prep_bwd only computes delta)
    if qkv.dtype == torch.bfloat16 and (_tl.attn_v3 & 14) != 14:
        n = B * H * 64 * Sp
        buf = torch.empty((3, n), dtype=qkv.dtype, device=qkv.device)
        qt, kt, dot = buf[0], buf[1], buf[2]
    lib().call("mh_attn_prep_bwd", _p(qkv), _p(o), _p(dout), _p(delta), _p(qt), _p(kt), _p(dot), B, S, H, dt(qkv), _stream())
    lib().call("mh_attn_bwd", _p(qkv), _p(dout), _p(lse), _p(delta), _p(qt), _p(kt), _p(dot), _p(dqkv), B, S, H, scale,
               _p(cos_t), _p(sin_t), dt(qkv), _stream())
    return dqkv


def tokattn_fwd(qkv, o, N: int, T: int, H: int, scale: float, cos_t=None, sin_t=None):
    """cos_t/sin_t: RoPE fused in (qkv unrotated); see mh_tokattn_fwd"""
    lib().call("mh_tokattn_fwd", _p(qkv), _p(o), N, T, H, scale, _p(cos_t), _p(sin_t), dt(qkv), _stream())
    return o


def tokattn_bwd(qkv, dout, dqkv, N: int, T: int, H: int, scale: float, cos_t=None, sin_t=None, rowscale=None):
    """``rowscale`` (fp32 [N * T]): row m of dqkv times rowscale[m] in the stores (the folded RMSNorm's d z)"""
    if rowscale is not None:
        assert rowscale.shape == (N * T,) and rowscale.dtype == torch.float32 and rowscale.is_contiguous()
        lib().call("mh_tokattn_bwd_scaled", _p(qkv), _p(dout), _p(dqkv), _p(rowscale), N, T, H, scale, _p(cos_t), _p(sin_t),
                   dt(qkv), _stream())
        return dqkv
    lib().call("mh_tokattn_bwd", _p(qkv), _p(dout), _p(dqkv), N, T, H, scale, _p(cos_t), _p(sin_t), dt(qkv), _stream())
    return dqkv


# -------------------------------------------------------------------------------------------- SwiGLU
def swiglu_fwd(gu, a):
    M, I2 = gu.shape
    lib().call("mh_swiglu_fwd", _p(gu), _p(a), M, I2 // 2, dt(gu), _stream())
    return a


def swiglu_bwd(gu, da, dgu):
    M, I2 = gu.shape
    lib().call("mh_swiglu_bwd", _p(gu), _p(da), _p(dgu), M, I2 // 2, dt(gu), _stream())
    return dgu


# ---------------------------------------------------------------------------------------------- loss
def cross_entropy(logits, V: int, target, row_loss, dlogits=None, scale_dev=None, argmax_out=None, ignore: int = 0):
    R = logits.shape[0]
    lib().call("mh_cross_entropy", _p(logits), _rowmajor(logits), _p(target), _p(row_loss), _p(dlogits), _p(scale_dev),
               _p(argmax_out), R, V, ignore, dt(logits), _stream())


def sum_f32(x, out):
    lib().call("mh_sum_f32", _p(x), x.numel(), _p(out), _stream())


def count_valid(target, ignore: int, count, inv):
    lib().call("mh_count_valid", _p(target), target.numel(), ignore, _p(count), _p(inv), _stream())


# ----------------------------------------------------------------------------------------- optimiser
def sumsq(g, partial1024, out, accumulate: bool):
    lib().call("mh_sumsq", _p(g), g.numel(), _p(partial1024), _p(out), int(accumulate), dt(g), _stream())


def clip_coef(sumsq_t, max_norm: float, coef, norm):
    lib().call("mh_clip_coef", _p(sumsq_t), max_norm, _p(coef), _p(norm), _stream())


def adamw(p, g, m, v, lr, b1, b2, eps, wd, bc1, bc2, coef_dev):
    lib().call("mh_adamw", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, wd, bc1, bc2, _p(coef_dev), dt(p), _stream())


# -------------------------------------------------------------------------------------------- decode
def kv_append(qkv, cos_t, sin_t, kc, vc, B: int, H: int, hd: int, Lmax: int, pos: int, pos_dev=None):
    lib().call("mh_kv_append", _p(qkv), _p(cos_t), _p(sin_t), _p(kc), _p(vc), B, H, hd, Lmax, pos, _p(pos_dev), dt(qkv),
               _stream())


def attn_decode(qkv, kc, vc, o, B: int, H: int, hd: int, Lmax: int, length: int, scale: float, pos_dev=None):
    lib().call("mh_attn_decode", _p(qkv), _p(kc), _p(vc), _p(o), B, H, hd, Lmax, length, scale, _p(pos_dev), dt(qkv),
               _stream())
    return o


def attn_decode_append(qkv, cos_t, sin_t, kc, vc, o, B: int, H: int, hd: int, Lmax: int, pos: int, scale: float, pos_dev=None):
    """kv_append + attn_decode in one launch (qkv is read unrotated and left untouched)"""
    lib().call("mh_attn_decode_append", _p(qkv), _p(cos_t), _p(sin_t), _p(kc), _p(vc), _p(o), B, H, hd, Lmax, pos, scale,
               _p(pos_dev), dt(qkv), _stream())
    return o


def kv_store_rows(qkv, kc, vc, B: int, S: int, H: int, hd: int, Lmax: int, pos0: int):
    """the rotated K, V of a chunk of S positions -> cache rows [pos0, pos0 + S)"""
    lib().call("mh_kv_store_rows", _p(qkv), _p(kc), _p(vc), B, S, H, hd, Lmax, pos0, dt(qkv), _stream())


def kv_gather_rows(kc, vc, qkv, B: int, n: int, Stot: int, H: int, hd: int, Lmax: int):
    """cache rows [0, n) -> K, V columns of rows [b*Stot, b*Stot + n) of qkv [B*Stot, 3*H*hd] (their q columns zeroed)"""
    lib().call("mh_kv_gather_rows", _p(kc), _p(vc), _p(qkv), B, n, Stot, H, hd, Lmax, dt(qkv), _stream())


def attn_fwd_tail(qkv, o, lse, B: int, S: int, H: int, scale: float, q_start: int):
    """attn_fwd computing only the query rows >= q_start (whole query tiles)"""
    lib().call("mh_attn_fwd_tail", _p(qkv), _p(o), _p(lse), B, S, H, scale, q_start, dt(qkv), _stream())
    return o


def kv_store_prefill(qkv, kc, vc, B: int, S: int, H: int, hd: int, Lmax: int):
    lib().call("mh_kv_store_prefill", _p(qkv), _p(kc), _p(vc), B, S, H, hd, Lmax, dt(qkv), _stream())


SAMPLE_MAX_K = 64


SAMPLE_MAX_RANGE = 2048
_ZERO_MASKS: dict = {}


def mask_spans(first_mask: torch.Tensor, lo_tab: torch.Tensor, hi_tab: torch.Tensor):
    """((first_lo, first_hi), [longest range per position]) of a grammar, computed once on the host"""
    nz = first_mask.nonzero().flatten()
    span = (int(nz.min()), int(nz.max()) + 1) if nz.numel() else (0, 0)
    return span, [int(x) for x in (hi_tab - lo_tab).max(dim=0).values.tolist()]


def sample_top_p_k(logits, first_mask, lo_tab, hi_tab, ev, pos: int, q, out, V: int, temp: float, top_p: float,
                   top_k: int, out_b=None, out_c=None, first_span=(0, 0), max_range: int = 0, ban_mask=None,
                   fill_rest: int = 0, fill_id: int = 0):
    """fused grammar-masked softmax + top-p/top-k + draw of token position `pos`; q [B, V] fp32 Exp(1) noise; the id goes
    to `out` (a strided int64 view [B]) and to the contiguous int64 [B] tensors out_b / out_c when given.  first_span =
    [lo, hi) outside of which first_mask is zero; max_range = the longest [lo_tab, hi_tab) range at this position
    (host-side facts about the tables, see mask_spans)"""
    B = logits.shape[0]
    assert q.dtype == torch.float32 and q.is_contiguous() and q.shape == (B, V) and out.dtype == torch.int64
    assert lo_tab.dtype == torch.int32 and lo_tab.is_contiguous() and hi_tab.is_contiguous() and ev.dtype == torch.int64
    for t in (out_b, out_c):
        assert t is None or (t.dtype == torch.int64 and t.is_contiguous() and t.numel() == B)
    if ban_mask is None:  # (the kernel reads the mask unconditionally: nothing banned = zeros)
        key = (logits.device, V)
        if key not in _ZERO_MASKS:
            _ZERO_MASKS[key] = torch.zeros(V, dtype=torch.uint8, device=logits.device)
        ban_mask = _ZERO_MASKS[key]
    lib().call("mh_sample_top_p_k", _p(logits), _rowmajor(logits), _p(first_mask), _p(ban_mask), int(first_span[0]),
               int(first_span[1]),
               _p(lo_tab), _p(hi_tab), lo_tab.shape[1], int(max_range), _p(ev), pos, _p(q), _p(out), out.stride(0),
               _p(out_b), _p(out_c), B, V, temp, top_p, top_k, fill_rest, fill_id, dt(logits), _stream())
    return out


def masked_softmax(logits, lo, hi, first_mask, probs, V: int, temp: float):
    B = logits.shape[0]
    lib().call("mh_masked_softmax", _p(logits), _rowmajor(logits), _p(lo), _p(hi), _p(first_mask), _p(probs), B, V, temp,
               dt(logits), _stream())
    return probs
