/* midihip.h — C-ABI of libmidihip.so: the MI355X (gfx950) kernels behind the midi-model hot path.
 *
 * The reference (SkyTNT/midi-model) has no FFI of its own: its hot path is Python that reaches
 * third-party native code through torch/transformers.  This header declares the
 * boundary a maintainer binds instead (ctypes stub in INTEGRATION.md); every entry point names the
 * reference call site it replaces.  TF: = transformers/ (5.15.0).
 *
 * Conventions
 *  - plain C: raw DEVICE pointers + sizes, no torch types; the caller owns all memory.
 *  - `dtype`: MH_F32 (parity/verification mode) or MH_BF16 (production); statistics (rstd, lse,
 *    losses, norms) and optimiser scalars are always fp32.
 *  - `stream` is a hipStream_t (NULL = default stream).  Calls only enqueue work.  The calling thread's CURRENT
 *    DEVICE must be the device `stream` and the buffers belong to (hipSetDevice before the call; torch keeps it so).
 *    The library holds no device memory of its own; what it caches (function attributes, one symbol address) is
 *    cached per device, so one process may drive several GPUs from several threads.
 *  - return 0 on success, <0 on error; mh_last_error() gives the message (thread-local).
 *  - token ids are int64 (torch.long), as on the reference API.
 *  - row-major everywhere; "ld*" are leading dimensions in ELEMENTS.
 */
#ifndef MIDIHIP_H
#define MIDIHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MH_F32 0
#define MH_BF16 1

#define MH_OK 0
#define MH_ERR_ARG (-1)
#define MH_ERR_LAUNCH (-2)
#define MH_ERR_UNSUPPORTED (-3)

const char* mh_last_error(void);
int mh_version(void);
/* 1 when the library was built with -DMH_AB_BUILDS (libmidihip_ab.so: the test / measurement library that also holds the
 * first-form attention kernels, the 128x128 bf16 GEMM and the ablation builds of the production GEMM), 0 for libmidihip.so. */
int mh_ab_builds(void);
/* runtime options (A/B runs and tests; every default is the production path; mh_get_option returns -1 for an unknown name).
 * THREAD-LOCAL: a value set here applies to later calls made by the SAME host thread only; other threads (e.g. the
 * concurrent generators of app.py:496) keep their own values, new threads start from the defaults.
 * "gemm" = 1 (production bf16 kernel: 256x256 tile, ping-pong wave groups) | 0 (128x128 two-stage kernel, the independent
 * check; bf16: A/B library only); "gemm_k64" = 1 (products with a row-major A operand -- forward projections, dgrads -- run
 * the K-step-64 main loop with whole-line LDS-DMA) | 2 (only row-major x row-major) | 0 (the K-step-32 loop everywhere;
 * identical bits); "gemm_ablate" = micro-benchmark / timeline builds of the production kernel (wrong results; A/B library
 * only -- the production library returns MH_ERR_ARG for any non-zero value at the next mh_gemm call); "gemm_lean_epi" = 1
 * (interior tiles take the lean forms of the plain / SwiGLU-backward epilogues: descriptor addressing, packed arithmetic) | 0
 * (the general forms everywhere; identical bits);
 * "skinny_mb" / "skinny_nbt" = 16-row activation blocks / 16-column blocks per workgroup of mh_gemm_skinny (0 = default);
 * "attn_v3" / "attn_v3_wps" = forms of the event-level attention kernels (attention_mfma3.hip; default 255: bit 7 = the
 * forward's lazy reference maximum; values that select a first-form kernel: A/B library only); "attn_passes" = in how many
 * chunks of tile ranks those kernels walk their (batch, head) pairs, light chunks last (default 5; 1 = pair after pair);
 * "tokattn_bwd_batched" = 1 (mh_tokattn_bwd at T = 8 reduces its dot products four per wave reduction) | 0 (one by one; the
 * same values up to fp32 summation order). */
int mh_set_option(const char* name, int value);
int mh_get_option(const char* name);

/* ---- dense projections (MFMA) ------------------------------------------------------------------
 * C[M,N] = alpha * A[M,K] * B[N,K]^T + beta * R[M,N]      (R may be NULL when beta == 0; R may alias C)
 * Replaces every nn.Linear of the path: q/k/v/o_proj (TF:models/llama/modeling_llama.py:254-256,280),
 * gate/up/down_proj (:174-176), lm_head (midi_model.py:107,135) and their autograd dgrad/wgrad.
 * K, lda, ldb must be multiples of 16 bytes worth of elements; M, N arbitrary.
 * `splitk` > 1 splits the contraction over grid.z: the kernel then only stores fp32 partial products in
 * `workspace` (splitk*M*N floats) and mh_gemm_splitk_reduce folds them into C (applying alpha/beta/R).   */
int mh_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R,
               int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int dtype, int splitk,
               void* workspace, void* stream);
/* General form: transA != 0 means A is stored contraction-major, A[K][M] (lda >= M); likewise transB: B[K][N].
 * That is how dgrad (dX = dY * W, W[N][K] contracted over its rows) and wgrad (dW = dY^T * X, both contracted
 * over the row index) present their operands, so neither needs a re-layout pass: the kernel stages such tiles as
 * they lie and assembles MFMA fragments with LDS transpose reads (ds_read_b64_tr_b16).  bf16 only; K is then
 * unrestricted.  For non-transposed operands whose K is not a multiple of 8 the row tail up to the next
 * multiple must be readable and finite (zero).                                                              */
int mh_gemm(const void* A, int64_t lda, int transA, const void* B, int64_t ldb, int transB, void* C, int64_t ldc,
            const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int dtype, int splitk,
            void* workspace, void* stream);
/* gate|up projection with the SwiGLU forward as its epilogue (LlamaMLP.forward, modeling_llama.py:174-176):
 *   GU[M, 2I] = A[M,K] * W[2I,K]^T (W = [gate_proj.weight; up_proj.weight], kept for the backward) and
 *   ACT[M, I] = round(silu(GU[:, :I])) * GU[:, I:]   -- the results of mh_gemm_nt followed by mh_swiglu_fwd, without
 * reading GU back.  GU == NULL: forward only (nothing will backpropagate: a prompt prefill), only ACT is written.
 * bf16, production GEMM kernel only (option "gemm" != 0), I % 128 == 0; fails loudly otherwise. */
int mh_gemm_swiglu(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT,
                   int64_t ldact, int64_t M, int64_t I, int64_t K, int dtype, void* stream);
/* q|k|v projection with the rotary embedding as its epilogue (LlamaAttention.forward: q_proj / k_proj / v_proj, then
 * apply_rotary_pos_emb on q and k, modeling_llama.py:151-169, 243-260): C[M, N = 3 * heads * 64] = A[M,K] * W[N,K]^T rounded
 * to bf16, then the q and k heads rotated at position pos0 + m % S.  `table` = bf16 [npos][96]: cos(32) | -sin(32) | +sin(32)
 * of pos * theta^(-2i/64), i.e. cos / sin already cast to the activation dtype as the reference does (modeling_llama.py:126).
 * Same results as mh_gemm followed by mh_rope, without the extra pass over q and k.  bf16, production GEMM kernel only
 * (option "gemm" != 0), head_dim 64; fails loudly otherwise. */
int mh_gemm_rope(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                 int64_t npos, int64_t S, int64_t pos0, int head_dim, int64_t M, int64_t N, int64_t K, int dtype, void* stream);
/* down_proj dgrad with the SwiGLU backward as its epilogue (LlamaMLP backward, modeling_llama.py:174-176):
 *   d a = A[M,K] * B[K,I] (B contraction-major, as mh_gemm with transB), rounded to the activation dtype, then
 *   DGU[:, :I] = d a * up * silu'(gate),  DGU[:, I:] = d a * silu(gate)     with GU = gate|up of the forward [M, 2I].
 * Same results as mh_gemm followed by mh_swiglu_bwd, without d a's round trip through HBM and the extra read of GU's
 * neighbourhood.  bf16, production GEMM kernel only (option "gemm" != 0), I % 8 == 0; fails loudly otherwise.   */
int mh_gemm_dswiglu(const void* A, int64_t lda, const void* B, int64_t ldb, const void* GU, int64_t ldgu, void* DGU,
                    int64_t lddgu, int64_t M, int64_t I, int64_t K, int dtype, void* stream);
/* ---- RMSNorm folded around the projections of a FORWARD-ONLY block (r05; LlamaRMSNorm + nn.Linear,
 * TF:models/llama/modeling_llama.py:62-67 with :254-256 / :174-176; the form mh_gemm_skinny's norm_eps runs for decode, at prefill
 * sizes).  With W' = w (.) W folded by the caller, norm(x) W^T = rstd (.) (x W'^T): the normalised activation is never written.
 *   mh_gemm_rowss        C = A B^T + R (R may be NULL) as mh_gemm_nt does, and rowss[(n / 64) * M + m] (fp32, N / 64 x M) = the sum
 *                        of squares of the STORED bf16 values C[m, 64 (n/64) .. +64): the statistics of the norm that follows, from
 *                        the registers the line is stored from.  N a multiple of 64; 16-byte aligned, ld multiples of 8.
 *   mh_row_rstd          rstd[m] = rsqrt(mean_k(row m ^2) + eps), from `parts` (= rowss, nparts = N / 64) or from the rows `x`
 *                        themselves (give exactly one of the two).
 *   mh_gemm_rope_scaled / mh_gemm_swiglu_scaled   mh_gemm_rope / mh_gemm_swiglu with every row of the product multiplied by
 *                        rowscale[m] (fp32 [M], 16-byte aligned, M a multiple of 4) before the epilogue's own arithmetic.   */
int mh_gemm_rowss(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R, int64_t ldr,
                  float* rowss, int64_t M, int64_t N, int64_t K, int dtype, void* stream);
int mh_row_rstd(const void* x, int64_t ldx, const float* parts, int nparts, int64_t M, int D, float eps, float* rstd, int dtype,
                void* stream);
int mh_gemm_rope_scaled(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                        int64_t npos, int64_t S, int64_t pos0, int head_dim, const float* rowscale, int64_t M, int64_t N, int64_t K,
                        int dtype, void* stream);
int mh_gemm_swiglu_scaled(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT, int64_t ldact,
                          const float* rowscale, int64_t M, int64_t I, int64_t K, int dtype, void* stream);
/* ---- the TRAINING form of the folded RMSNorm (r06; engine.layer_forward_train_folded / layer_backward_folded) ----------------------
 * Forward: the q|k|v and gate|up projections take the residual stream x itself, multiply by W' = W (.) w (mh_scale_cols) and scale
 * their rows by rstd (mh_gemm_rope_scaled / mh_gemm_swiglu_scaled / mh_gemm_nt_scaled); rstd comes from the producing projection's
 * statistics (mh_gemm_rowss + mh_row_rstd): the normalised activations are never written, read or kept for the backward.
 * Backward: the producers of the projections' output gradients store d z = rstd (.) d y (mh_gemm_dswiglu_scaled, mh_attn_bwd_o_scaled,
 * mh_tokattn_bwd_scaled); t = d z W' (mh_gemm, the folded weights, contraction-major); dx = t - x (rstd^2 / D) rowdot(t, x) + dres
 * (mh_rmsnorm_bwd_folded); the weight gradient G' = d z^T x is reduced by mh_gemm_splitk_reduce_fold, which applies the chain rule
 * through the fold: dW = G' (.) w (+ beta R) and per-block partial column sums of dw = colsum(G' (.) W) in `colpart`
 * [mh_splitk_fold_blocks(M), N] fp32 (fold them with mh_colsum).  bf16, production GEMM kernel. */
int mh_gemm_nt_scaled(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* rowscale, int64_t M,
                      int64_t N, int64_t K, int dtype, void* stream);
int mh_gemm_dswiglu_scaled(const void* A, int64_t lda, const void* B, int64_t ldb, const void* GU, int64_t ldgu, void* DGU,
                           int64_t lddgu, const float* rowscale, int64_t M, int64_t I, int64_t K, int dtype, void* stream);
int mh_splitk_fold_blocks(int64_t M);
int mh_gemm_splitk_reduce_fold(const void* workspace, void* C, int64_t ldc, const void* R, int64_t ldr, int64_t M, int64_t N,
                               int splitk, float alpha, float beta, const void* wnorm, const void* W, int64_t ldw, float* colpart,
                               int dtype, void* stream);
int mh_rmsnorm_bwd_folded(const void* x, const float* rstd, const void* t, const void* dres, void* dx, int64_t M, int D, int dtype,
                          void* stream);
int mh_scale_cols(const void* W, int64_t ldw, const void* w, void* out, int64_t ldo, int64_t Nr, int K, int dtype, void* stream);
/* mh_scale_cols for a list of contiguous [rows, K] matrices in ONE launch: jobs (device memory) = njobs x {W, w, out, rows} as int64 */
int mh_scale_cols_batched(const int64_t* jobs, int njobs, int K, int dtype, void* stream);
int mh_gemm_splitk_reduce(const void* workspace, void* C, int64_t ldc, const void* R, int64_t ldr, int64_t M,
                          int64_t N, int splitk, float alpha, float beta, int dtype, void* stream);
/* Skinny projection of the decode step (replaces the per-token nn.Linear calls of LlamaAttention / LlamaMLP /
 * lm_head when q_len == 1, modeling_llama.py:243-281, 174-176; midi_model.py:135): 1 <= M <= 64 rows, bf16.
 *   MH_SKINNY_PLAIN   C[M,N] = A[M,K] * W[N,K]^T (+ R)
 *   MH_SKINNY_GATEUP  W = [gate; up] (2N rows): C[M,N] = round(silu(round(A gate^T))) * round(A up^T)   (LlamaMLP)
 * K a multiple of 256.  norm_eps > 0: every row of the product is scaled by rsqrt(mean_k A[m,k]^2 + norm_eps) before
 * the epilogue -- with the RMSNorm weight folded into W by the caller this is LlamaRMSNorm (:62-67) + projection.
 * row_ids / res_ids (optional, int64 [M]): row m of A / of R is row ids[m] of the given table (nn.Embedding lookup of
 * the token just sampled, midi_model.py:126-131, without a launch of its own).
 * LIMITS (32-bit element offsets inside the kernel): N * ldw (2 N * ldw for GATEUP), 64 * lda, 64 * ldc, 64 * ldr < 2^31 are
 * checked by the entry point; with row_ids / res_ids the ids live in device memory, so the CALLER guarantees
 * (max id + 1) * lda < 2^31 and (max id + 1) * ldr < 2^31 elements (a gathered table of at most 2^31 - 1 bf16 elements = 4 GiB;
 * this model's embedding tables hold 3.5 M).  A larger table needs the lookup done by the caller (mh_copy_rows). */
#define MH_SKINNY_PLAIN 0
#define MH_SKINNY_GATEUP 1
int mh_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* R,
                   int64_t ldr, int mode, float norm_eps, const int64_t* row_ids, const int64_t* res_ids, int64_t M,
                   int64_t N, int64_t K, int dtype, void* stream);
/* out[C,R] = in[R,C]^T (operand re-layout for dgrad/wgrad). */
int mh_transpose(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int64_t cols, int dtype,
                 void* stream);

/* ---- embeddings --------------------------------------------------------------------------------
 * out[m,:] = sum_j table[tok[m,j],:]   (midi_model.py:145-146: embed_tokens(x).sum(-2))             */
int mh_embed_sum_fwd(const int64_t* tok, const void* table, void* out, int64_t M, int T, int64_t V, int D,
                     int dtype, void* stream);
/* out[m,0,:] = hidden[m,:]; out[m,j,:] = table[tok[m,j-1],:], j=1..T-1   (midi_model.py:126-131)   */
int mh_concat_tok_fwd(const void* hidden, const int64_t* tok, int64_t ldtok, const void* table, void* out,
                      int64_t M, int T, int64_t V, int D, int dtype, void* stream);
/* dtable_f32[tok[m,j],:] += dout_row(m,j): scatter-add of embedding gradients into an fp32 [V,D] accumulator;
 * rows with tok == pad_id are skipped (nn.Embedding padding_idx, TF:models/llama/modeling_llama.py:353).
 * dout_row(m,j) starts at dout + (m*rows_per_m + j*jstride + j0)*D: (1,0,0) for the summed event embedding,
 * (T,1,1) for the token-sequence embedding whose row 0 is the hidden state.                              */
int mh_embed_scatter_bwd(const int64_t* tok, int64_t ldtok, int T, const void* dout, int rows_per_m, int jstride,
                         int j0, float* dtable_f32, int64_t M, int64_t V, int D, int64_t pad_id, int dtype,
                         void* stream);
/* Segment form (production): occurrences pre-sorted by token id.  dtable_f32[v,:] += sum_{i in [seg_start[v],
 * seg_start[v+1])} dout[src_rows[i]*ld ...]; id `pad_id` is skipped.  n_occ = seg_start[V] = length of src_rows. */
int mh_embed_segment_bwd(const int64_t* src_rows, const int64_t* seg_start, const void* dout, int64_t ld,
                         float* dtable_f32, int64_t V, int D, int64_t n_occ, int64_t pad_id, int dtype, void* stream);
/* Index preparation of mh_embed_segment_bwd: a counting sort of the occurrences of the id matrix tok[n_rows, n_cols] (row
 * stride ldtok) by token id (replaces the reference's implicit scatter of nn.Embedding's backward, TF:models/llama/
 * modeling_llama.py:353; midi_model.py:126-131,143-145 are the two embeddings).  seg_start[v] (V + 1 entries) = occurrences
 * with id < v; src_rows[p] = r * row_mul + j * col_mul + add for the occurrence (r, j) placed at p -- the row of the gradient
 * matrix it reads: (1, 0, 0) for the summed event embedding, (T, 1, 1) for the token-sequence embedding.  Ids outside [0, V)
 * are a caller error (nn.Embedding raises); they are grouped behind seg_start[V], nothing is written out of bounds.
 * `work`: int32 scratch of (ceil(n_rows * n_cols / mh_token_segments_chunk()) + 1) * (V + 1) elements.  The order inside a
 * segment is unspecified. */
int mh_token_segments_chunk(void);
int mh_token_segments(const int64_t* tok, int64_t ldtok, int64_t n_rows, int n_cols, int64_t V, int64_t row_mul,
                      int64_t col_mul, int64_t add, int64_t* src_rows, int64_t* seg_start, int32_t* work, void* stream);
/* dst[i] (dtype) = (accumulate ? dst[i] : 0) + src_f32[i] */
int mh_cast_from_f32(const float* src, void* dst, int64_t n, int accumulate, int dtype, void* stream);
/* strided row copy: dst[m,:] = src[m*src_ld ...] (+ optional accumulate) — takes d(hidden) out of d(token seq). */
int mh_copy_rows(const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int64_t M, int D, int accumulate,
                 int dtype, void* stream);

/* Batch assembly (MidiDataset.__getitem__ slicing + collate_fn, train.py:69-90) from a corpus resident in device memory:
 * tokens int16 [n_events, T]; window b = events [win_start[b], win_start[b] + win_len[b]) (the caller keeps windows
 * inside the corpus); out int64 [B, L, T], positions past a window's length filled with pad_id.                */
int mh_collate_windows(const int16_t* tokens, int64_t n_events, const int64_t* win_start, const int64_t* win_len,
                       int64_t* out, int64_t B, int64_t L, int T, int64_t pad_id, void* stream);

/* Data augmentation fused into the batch assembly: MIDITokenizer.augment (midi_tokenizer.py:364-417 v1, :1023-1102 v2), which
 * MidiDataset.load_midi applies to every file it serves (train.py:62-63, aug=True by default).  Integer work: bit-exact.
 * `tab` int32[40]: the tokenizer tables the rules need (midi-model_amd/tokenizer.py: augment_table; layout in csrc/augment.hip).
 * mh_augment_piece_stats -- once per corpus: for every piece (file) p = events [piece_off[p], piece_off[p+1]) of the int16
 *   corpus, stats[p*130 ...] = {lowest, highest pitch among its notes off the drum channel (128, -1 when there are none),
 *   bit mask of the ORIGINAL channels of the notes of each ORIGINAL track [128]}: the whole-file facts behind the reference's
 *   "file unchanged when a shifted pitch leaves 0..127" (:1065-1066) and its key-signature second pass (:1099-1104).
 * mh_augment_collate_windows -- per batch: mh_collate_windows with window b taken from piece win_piece[b] and augmented with
 *   shifts[b*6 ...] = {pitch, velocity, cc value, bpm, track, channel} (the reference's draw order, :1025-1030).          */
int mh_augment_piece_stats(const int16_t* tokens, const int64_t* piece_off, int64_t n_pieces, const int32_t* tab,
                           int32_t* stats, void* stream);
int mh_augment_collate_windows(const int16_t* tokens, int64_t n_events, const int64_t* win_start, const int64_t* win_len,
                               const int64_t* win_piece, const int32_t* shifts, const int32_t* stats, const int32_t* tab,
                               int64_t* out, int64_t B, int64_t L, int T, int64_t pad_id, void* stream);

/* ---- RMSNorm (TF:models/llama/modeling_llama.py:62-67) ---------------------------------------------
 * y = w * T(x * rsqrt(mean(x^2)+eps));  rstd[M] (fp32) is saved for the backward.                   */
int mh_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t M, int D, float eps, int dtype,
                   void* stream);
/* dx = rmsnorm'(dy) (+ dres if non-NULL: the residual-stream gradient);  dw_partial[nblk,D] fp32 gets the
 * per-block column sums of dy * xhat (reduce with mh_colsum).  nblk = mh_rmsnorm_bwd_blocks(M).        */
int mh_rmsnorm_bwd_blocks(int64_t M);
int mh_rmsnorm_bwd(const void* x, const void* w, const float* rstd, const void* dy, const void* dres, void* dx,
                   float* dw_partial, int64_t M, int D, int dtype, void* stream);
int mh_colsum(float* partial /* clobbered */, int64_t nblk, void* out, int D, int accumulate, int dtype, void* stream);

/* ---- RoPE (TF:models/llama/modeling_llama.py:113-160; half-split rotate_half) --------------------------
 * In place on the q and k thirds of qkv[M, 3*H*hd]; row m sits at position pos0 + (m % S).
 * cos/sin: fp32 tables [npos, hd/2].  dir=+1 forward, -1 backward (transpose rotation).               */
int mh_rope(void* qkv, const float* cos_t, const float* sin_t, int64_t M, int64_t S, int64_t pos0, int H, int hd,
            int dir, int dtype, void* stream);

/* ---- event-level causal attention, head_dim 64 (TF:integrations/sdpa_attention.py:79-166) -------------
 * qkv[B*S, 3*H*64] (q|k|v thirds, RoPE already applied) -> o[B*S, H*64], lse[B,H,S] (natural log).
 * bf16: MFMA flash kernel, needs the transposed copy vt[B,H,64,Sp] (Sp = S rounded up to 64, padding
 * zero) written by mh_attn_prep_fwd.  fp32: plain verification kernel (vt ignored).                 */
int mh_attn_prep_fwd(const void* qkv, void* vt, int64_t B, int64_t S, int H, int dtype, void* stream);
int mh_attn_fwd(const void* qkv, const void* vt, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                int dtype, void* stream);
/* backward: dqkv[B*S,3*H*64] <- (qkv, o, do, lse).  scratch: delta[B,H,S] fp32; bf16 additionally
 * qt,kt,dot: [B,H,64,Sp] transposed copies (zero padded) filled by mh_attn_prep_bwd.
 * cos_t/sin_t (optional, fp32 [>=S, 32]): the q and k thirds of dqkv are returned rotated back (the transpose of
 * apply_rotary_pos_emb at position = row index within the sequence), i.e. the gradient with respect to the
 * unrotated projection output -- what mh_rope(dqkv, dir = -1) would make of it in a pass of its own.            */
int mh_attn_prep_bwd(const void* qkv, const void* o, const void* dout, float* delta, void* qt, void* kt, void* dot,
                     int64_t B, int64_t S, int H, int dtype, void* stream);
int mh_attn_bwd(const void* qkv, const void* dout, const float* lse, const float* delta, const void* qt,
                const void* kt, const void* dot, void* dqkv, int64_t B, int64_t S, int H, float scale,
                const float* cos_t, const float* sin_t, int dtype, void* stream);
/* the same backward in ONE call for the default bf16 kernels (third form with transpose reads, no transposed copies): the dQ
 * kernel computes delta = rowsum(dO * O) from the rows it holds anyway and leaves it in `delta` for the dK/dV kernel launched
 * behind it -- no pass of its own over O and dO.  `delta` is scratch of 2 * B*H*Sp floats here: the second half receives
 * -lse * log2(e), the form the dK/dV kernel's exp2 wants (one multiply less per score).  MH_ERR_ARG for fp32 or when another kernel form
 * is selected (mh_set_option("attn_v3")): use the two calls above.                                                      */
int mh_attn_bwd_o(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int64_t B,
                  int64_t S, int H, float scale, const float* cos_t, const float* sin_t, int dtype, void* stream);
/* mh_attn_bwd_o with row m = b * S + position of dqkv multiplied by rowscale[m] in the kernels' stores (the folded RMSNorm's d z) */
int mh_attn_bwd_o_scaled(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                         const float* rowscale, int64_t B, int64_t S, int H, float scale, const float* cos_t, const float* sin_t,
                         int dtype, void* stream);
/* measurement aid, A/B library only (the production library returns MH_ERR_UNSUPPORTED): the production bf16 forward with shader-clock
 * stamps at the seams of each key tile's segments; stamps: uint32 [16][4][32][9] (tools/attn_timeline.py decodes them).        */
int mh_attn_fwd_timeline(const void* qkv, void* o, float* lse, int64_t B, int64_t S, int H, float scale, int lazy,
                         uint32_t* stamps, void* stream);
/* NOTE: lse and delta are laid out [B,H,Sp] with Sp = S rounded up to a multiple of 64 (entries past S unused).
 * The *_plain variants run the exact-fp32-math thread-per-row kernels for either dtype; tests use them to
 * cross-check the MFMA kernels on the device.                                                              */
int mh_attn_fwd_plain(const void* qkv, void* o, float* lse, int64_t B, int64_t S, int H, float scale, int dtype,
                      void* stream);
int mh_attn_bwd_plain(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int64_t B,
                      int64_t S, int H, float scale, int dtype, void* stream);

/* ---- token-level attention: sequences of T<=8 tokens, head_dim 256 (net_token) --------------------------
 * qkv[N*T, 3*H*256] -> o[N*T, H*256], causal within each sequence of T rows.
 * cos_t/sin_t (optional, fp32 [>=T, 128]): apply_rotary_pos_emb (modeling_llama.py:130-160) fused in -- q,k are read
 * unrotated and rotated in registers at position = token index; the backward then returns the gradient with respect
 * to the unrotated q,k.  Null: q,k are taken as they are (rotated beforehand with mh_rope).                  */
int mh_tokattn_fwd(const void* qkv, void* o, int64_t N, int T, int H, float scale, const float* cos_t, const float* sin_t,
                   int dtype, void* stream);
int mh_tokattn_bwd(const void* qkv, const void* dout, void* dqkv, int64_t N, int T, int H, float scale, const float* cos_t,
                   const float* sin_t, int dtype, void* stream);
/* the same with row m = n * T + t of dqkv multiplied by rowscale[m] in the stores (the folded RMSNorm's d z) */
int mh_tokattn_bwd_scaled(const void* qkv, const void* dout, void* dqkv, const float* rowscale, int64_t N, int T, int H, float scale,
                          const float* cos_t, const float* sin_t, int dtype, void* stream);

/* ---- SwiGLU (TF:models/llama/modeling_llama.py:174-176) -------------------------------------------------
 * gu[M,2I] = [gate | up];  a = silu(gate) * up;  dgu = [da*up*silu'(gate) | da*silu(gate)]            */
int mh_swiglu_fwd(const void* gu, void* a, int64_t M, int I, int dtype, void* stream);
int mh_swiglu_bwd(const void* gu, const void* da, void* dgu, int64_t M, int I, int dtype, void* stream);

/* ---- cross-entropy over the vocabulary (train.py:180-185) ------------------------------------------------
 * logits[R, ldl] (V valid columns).  row_loss[r] = lse - logit[target] (0 when target == ignore).
 * If dlogits != NULL: dlogits = (softmax - onehot) * (*scale_dev) for kept rows, 0 for ignored rows and for
 * padding columns V..ldl-1 (dlogits may alias logits).  argmax_out (optional) gets the row argmax.       */
int mh_cross_entropy(const void* logits, int64_t ldl, const int64_t* target, float* row_loss, void* dlogits,
                     const float* scale_dev, int64_t* argmax_out, int64_t R, int V, int64_t ignore, int dtype,
                     void* stream);
/* out[0] = sum(x[0..n)) (deterministic tree); out[1] = number of x != 0 is NOT computed here.             */
int mh_sum_f32(const float* x, int64_t n, float* out, void* stream);
/* count[0] = #(target != ignore); inv[0] = 1/max(count,1)                                                 */
int mh_count_valid(const int64_t* target, int64_t n, int64_t ignore, float* count, float* inv, void* stream);

/* ---- optimiser (train.py:121-151 AdamW, Trainer gradient_clip_val=1.0 train.py:464) -----------------------
 * mh_sumsq: out[0] = (accumulate ? out[0] : 0) + sum(g^2), deterministic two-stage reduction through the
 * caller's 1024-float scratch `partial1024`.  mh_clip_coef: coef[0] = min(1, max_norm/(sqrt(sumsq)+1e-6)),
 * norm[0] = sqrt(sumsq).  mh_adamw: torch.optim.AdamW single-tensor update with g scaled by *coef_dev. */
int mh_sumsq(const void* g, int64_t n, float* partial1024, float* out, int accumulate, int dtype, void* stream);
int mh_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm, void* stream);
int mh_adamw(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, float bias_corr1, float bias_corr2, const float* coef_dev, int dtype, void* stream);

/* ---- KV-cached single-event decode (midi_model.py:195-246; TF:cache_utils.py:127-147) -----------------------
 * Cache layout per layer: k,v [B,H,Lmax,hd].  mh_kv_append: rotate q,k of qkv[B,3*H*hd] at position `pos`
 * in place and store k,v rows at index pos.  mh_attn_decode: o[B,H*hd] = softmax(q K^T * scale) V over
 * the first `len` cached rows (q_len == 1: no causal mask, sdpa_attention.py:120).  hd in {64,256}.
 * pos_dev (optional, device int32): when non-null the kernels read the position from it at run time -- row index
 * *pos_dev for the append, rows [0, *pos_dev] for the attention -- so that a captured hipGraph of one decode step
 * can be replayed for every event; `pos` / `len` are then ignored.                                         */
int mh_kv_append(void* qkv, const float* cos_t, const float* sin_t, void* kcache, void* vcache, int64_t B, int H,
                 int hd, int64_t Lmax, int64_t pos, const int32_t* pos_dev, int dtype, void* stream);
int mh_attn_decode(const void* qkv, const void* kcache, const void* vcache, void* o, int64_t B, int H, int hd,
                   int64_t Lmax, int64_t len, float scale, const int32_t* pos_dev, int dtype, void* stream);
/* Both in one launch: qkv holds the unrotated q,k,v of position `pos` (or *pos_dev); every (b,h) block stores its rotated
 * k row and v row in the cache, rotates q in registers and attends over rows [0, pos].  qkv is left untouched.        */
int mh_attn_decode_append(const void* qkv, const float* cos_t, const float* sin_t, void* kcache, void* vcache, void* o,
                          int64_t B, int H, int hd, int64_t Lmax, int64_t pos, float scale, const int32_t* pos_dev,
                          int dtype, void* stream);
/* A chunk of q_len > 1 new positions behind n cached ones (a cache-carrying forward, midi_model.py:137-150 with a non-empty
 * DynamicCache; TF:integrations/sdpa_attention.py:79-166): mh_kv_store_rows appends the chunk's rotated K and V at cache rows
 * [pos0, pos0 + S); mh_kv_gather_rows copies cache rows [0, n) into the K and V columns of rows [b*Stot, b*Stot + n) of a fused
 * qkv buffer [B*Stot, 3*H*hd] (q columns zeroed); mh_attn_fwd_tail is mh_attn_fwd over that buffer computing only the query
 * rows >= q_start (whole query tiles: rows of the first tile below q_start are written too and are not meaningful).        */
int mh_kv_store_rows(const void* qkv, void* kcache, void* vcache, int64_t B, int64_t S, int H, int hd, int64_t Lmax,
                     int64_t pos0, int dtype, void* stream);
int mh_kv_gather_rows(const void* kcache, const void* vcache, void* qkv, int64_t B, int64_t n, int64_t Stot, int H, int hd,
                      int64_t Lmax, int dtype, void* stream);
int mh_attn_fwd_tail(const void* qkv, void* o, float* lse, int64_t B, int64_t S, int H, float scale, int64_t q_start,
                     int dtype, void* stream);
/* copy rotated K and V of a prefill (qkv[B*S,3*H*hd]) into the cache rows [0,S).                           */
int mh_kv_store_prefill(const void* qkv, void* kcache, void* vcache, int64_t B, int64_t S, int H, int hd,
                        int64_t Lmax, int dtype, void* stream);

/* ---- grammar-masked softmax for the sampler (midi_model.py:202-223) -----------------------------------------
 * probs[b,:] = softmax(logits[b,:]/temp) * mask_b, mask_b = ids in [lo[b],hi[b]) or, when lo[b] < 0, the
 * `first_mask` table (event ids + EOS).  fp32 output, as torch.softmax on fp32 logits gives.              */
int mh_masked_softmax(const void* logits, int64_t ldl, const int32_t* lo, const int32_t* hi,
                      const uint8_t* first_mask, float* probs, int64_t B, int V, float temp, int dtype,
                      void* stream);
/* Fused sampler of one token position (midi_model.py:202-228, 152-165; torch.multinomial's single-draw path):
 * softmax(logits / temp) masked by the grammar -- position 0: `first_mask` (event ids + EOS); position pos >= 1: ids in
 * [lo_tab[e][pos], hi_tab[e][pos]) for the row's event id e = ev[b] (tables [*, tab_stride] int32) -- then keep the
 * top_k largest (value descending, index ascending), zero those whose preceding cumulative mass exceeds top_p,
 * renormalise, and return for every row the id maximising p_j / q[b, j], where q [B, V] holds Exp(1) draws (only the
 * first top_k of a row are read) taken by the caller from the caller's generator (torch.Tensor.exponential_), which is
 * what keeps a seeded generator's stream identical to the reference's.  The id goes to out[b * out_stride] and, when
 * non-null, to out_b[b] and out_c[b] (all int64); the fill_rest entries after out[b * out_stride] are set to fill_id
 * (position 0 opens a fresh event row padded with pad_id).  1 <= top_k <= 64.  ban_mask ([V] bytes, required -- all zero = nothing banned): ids with a
 * non-zero byte are removed from every mask (app.py:30-31,85-86 disable_channels).  The caller states the spans of the masks:
 * first_mask is zero outside [first_lo, first_hi), no table range AT THIS POSITION is longer than max_range; both at most
 * 2048 ids.                                                                                                 */
int mh_sample_top_p_k(const void* logits, int64_t ldl, const uint8_t* first_mask, const uint8_t* ban_mask, int first_lo,
                      int first_hi,
                      const int32_t* lo_tab, const int32_t* hi_tab, int tab_stride, int max_range, const int64_t* ev,
                      int pos, const float* q, int64_t* out,
                      int64_t out_stride, int64_t* out_b, int64_t* out_c, int64_t B, int V, float temp, float top_p,
                      int top_k, int fill_rest, int64_t fill_id, int dtype, void* stream);

/* ---- the data-parallel gradient exchange on RCCL (reference: DDP under Lightning, train.py:461-474) -----------------
 * One communicator per process/GPU.  librccl is opened on the first call (dlopen; not a link-time dependency).
 * mh_comm_unique_id: rank 0 fills 128 bytes (ncclGetUniqueId) that the host ships to every rank;
 * mh_comm_init: ncclCommInitRank on `device`; *comm_out is the handle;
 * mh_comm_allreduce: IN PLACE over buf[count] on `stream`; mean != 0 averages over the ranks inside the collective
 *   (pre-multiplied sum, scale 1/world) -- DDP's divide-then-sum as one pass; mean == 0 sums.  The scale is 1/world rounded
 *   to the buffer's dtype: exact for power-of-two worlds (1, 2, 4, 8 GPUs of a node); for other worlds a bf16 contribution is
 *   multiplied by bf16(1/world) where DDP divides by world -- a difference of one rounding of the scale;
 * mh_comm_broadcast: buf[count] from `root` to every rank, in place;  mh_comm_info: rank / world / RCCL version code.
 * mh_comm_init leaves the calling thread's current device as it found it.  Handles are tracked: a call on a handle that was
 * never returned by mh_comm_init, or was already destroyed, returns MH_ERR_ARG (it is not dereferenced).              */
int mh_comm_unique_id(void* id128);
int mh_comm_init(int rank, int world, const void* id128, int device, void** comm_out);
int mh_comm_info(void* comm, int* rank, int* world, int* rccl_version);
int mh_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, int mean, void* stream);
int mh_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream);
int mh_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
