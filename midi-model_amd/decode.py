"""KV-cached decode driver for ``MIDIModel.generate`` (midi_model.py:167-250).

The reference spends a generated event in ~1000 launches and B host syncs.  Here one event is THREE graph replays and
ONE device->host copy:

    noise graph  (side stream, overlapping the previous event's net step): the Exp(1) variates torch.multinomial would draw
                 inside sample_top_p_k for the 8 token positions, [8, B, vocab], from a session generator that carries the
                 caller's generator state in and out
    steps graph  all 8 token steps back to back: (embedding of the token just sampled ->) 3 decoder layers at position i ->
                 lm_head -> fused grammar-masked softmax + top-p/top-k + draw (mh_sample_top_p_k) on that noise
    [copy]       the event's 8 tokens to the host; the reference's break rule (midi_model.py:232-235) is evaluated on them
                 AFTERWARDS: positions past an event's arity sample PAD whether or not the loop runs them, so running all 8
                 changes nothing but the number of draws -- and the generator is wound back to exactly the draws the
                 reference would have made (philox offset arithmetic), so a seeded torch.Generator still yields the
                 reference loop's stream (tests/test_decode_gpu.py drives app.py's loop against generate())
    net graph    embedding sum of the event's 8 tokens -> 12 decoder layers on one position per sequence -> final norm;
                 K/V appended to preallocated caches at a position kept in DEVICE memory

Samplers the fused kernel does not cover (top_k > 64) and the eager mode (MH_DECODE_GRAPHS=0, the CPU fake backend of the
tests) keep the step-by-step form: one graph / call per token step with the reference's own sampling ops inside.

The graphs are hipGraphs captured through torch.cuda.CUDAGraph from the same Python schedule the eager path runs
(``engine.stack_decode``), so there is one implementation of the step.  A session owns every buffer the graphs touch and
is keyed by (parameter buffer, batch, capacity, temperature, top_p, top_k); ``MIDIModel`` keeps a pool of them and hands one to each
``generate`` call, so concurrent generators on one model (app.py:496) never share scratch.
"""
from __future__ import annotations

import os
import threading
from typing import List, Optional

import torch

from . import engine, ops
from .engine import KVState, RopeTable


# One capture at a time per process, and in thread-local error mode: generators of other threads (app.py runs up to 10 on
# one model) keep launching and allocating while a new session is being captured.
# Replays of graphs that carry a registered generator state (the noise graph; the per-step graphs of the unfused sampler) take the
# same lock: torch's replay prologue asserts that NO capture is active, and under HIP a capture on another thread's stream makes
# that check fire ("Cannot prepare for replay during capturing stage", seen once in ~3 runs of the whole GPU suite in
# test_concurrent_generators_on_one_model, r04).  Captures are rare (a session is captured once and pooled), so the replays of
# other generators wait a few hundred milliseconds at most.
_CAPTURE_LOCK = threading.RLock()
_NOISE_INLINE = os.environ.get("MH_DECODE_NOISE_INLINE", "0") == "1"
_COPY_PAGEABLE = os.environ.get("MH_DECODE_COPY_PAGEABLE", "0") == "1"


def graphs_enabled(device: torch.device) -> bool:
    return device.type == "cuda" and os.environ.get("MH_DECODE_GRAPHS", "1") != "0"


class DecodeSession:
    def __init__(self, model, B: int, capacity: int, temp: float, top_p: float, top_k: int):
        tok = model.tokenizer
        self.model, self.B, self.cap, self.temp = model, B, capacity, float(temp)
        self.top_p, self.top_k = float(top_p), int(top_k)
        self.key = self.make_key(model, B, capacity, temp, top_p, top_k)
        dev, dt = model.device, model.dtype
        self.T, self.V, self.Vp = tok.max_token_seq, tok.vocab_size, model.vocab_padded
        spec, tspec = model._specs["net"], model._specs["net_token"]
        self.kv1 = KVState(spec, B, capacity, model._flat)
        self.kv2 = KVState(tspec, B, self.T, model._flat)
        # rope tables private to the session: a captured graph keeps their addresses
        self.rope1 = RopeTable(spec.hd, spec.theta, dev, capacity)
        self.rope2 = RopeTable(tspec.hd, tspec.theta, dev, self.T)
        first, lo, hi, _ = model._grammar()
        self.first_mask = first.clone()           # generate() overwrites it (ban_eos, disable_patch/control_change)
        self.ban = torch.zeros_like(first)        # ids removed from every mask (disable_channels); generate() fills it
        self.lo_tab, self.hi_tab = lo, hi
        self.first_span, self.max_range = ops.mask_spans(first, lo, hi)  # (generate() only ever clears bits of first_mask)
        self.fused_sampler = (1 <= self.top_k <= min(ops.SAMPLE_MAX_K, self.V) and max(self.max_range) <= ops.SAMPLE_MAX_RANGE
                              and self.first_span[1] - self.first_span[0] <= ops.SAMPLE_MAX_RANGE)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)           # cached events so far (device side)
        self.hidden = torch.zeros((B, spec.D), dtype=dt, device=dev)
        self.seq = torch.zeros((B, self.T), dtype=torch.long, device=dev)  # tokens of the event being sampled
        self.samples_in = torch.zeros((B,), dtype=torch.long, device=dev)  # token sampled at the previous position
        self.ev = torch.zeros((B,), dtype=torch.long, device=dev)          # event id (token 0) of the current event
        self.pad_id = tok.pad_id
        self.neg1 = torch.full((B,), -1, dtype=torch.int32, device=dev)
        self.probs = torch.zeros((B, 1, self.V), dtype=torch.float32, device=dev)
        # Exp(1) noise of the sampler, one [B, V] slice per token position (ones: a step sampled before any draw is greedy-safe)
        self.q_all = torch.ones((self.T, B, self.V), dtype=torch.float32, device=dev)
        self.q = self.q_all[0]
        self.logits = torch.zeros((B, self.Vp), dtype=dt, device=dev)
        # RMSNorm weights folded into the projections that follow them (one launch for norm + projection)
        self.fold1 = self.fold2 = self.lm_fold = None
        probe = torch.empty((B, 1), dtype=dt, device=dev)
        if ops.skinny_ok(probe, spec.D) and ops.skinny_ok(probe, spec.I) and ops.skinny_ok(probe, tspec.I) \
                and os.environ.get("MH_DECODE_FOLD", "1") != "0":
            self.fold1 = engine.fold_norm_weights(model._W["net"])
            self.fold2 = engine.fold_norm_weights(model._W["net_token"])
            self.lm_fold = torch.empty_like(model.lm_head.weight.data)
            self.refresh()
        self.g_net = None
        self.g_tok: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.T
        self.g_noise = self.g_steps = None
        self.use_graphs = graphs_enabled(dev)
        self.gen = torch.Generator(device=dev) if self.use_graphs else None  # the generator the captured sampler draws from
        self._user_gen = None
        self._pool = None
        self._off = 0          # philox offset of the next draw the reference loop would make
        self._draw_inc = 0     # philox offset consumed by one [B, V] exponential_ call
        self._noise_pending = False
        self._noise_read = False   # the pending variates have been read by a tok_step (step-by-step form)
        self._steps_recorded = False
        self.noise_stream = self.noise_done = self.copy_stream = self.steps_done = self.seq_host = None
        if self.use_graphs:
            self._capture()

    @staticmethod
    def make_key(model, B, capacity, temp, top_p, top_k):
        return (model._flat.data_ptr(), model._flat.dtype, B, capacity, float(temp), float(top_p), int(top_k))

    def refresh(self) -> None:
        """Re-derive the folded weights (norm weight x projection) IN PLACE from the live parameters.  The training
        kernels write the flat parameter buffer through raw pointers (AdamW, LoRA materialisation), which no tensor version
        counter sees, so a pooled session refreshes every time it is handed out (model._checkout_session): ~0.3 GB of
        elementwise traffic per generate() call, and the captured graphs keep their addresses."""
        if self.fold1 is None:
            return
        m = self.model
        engine.fold_norm_weights(m._W["net"], out=self.fold1)
        engine.fold_norm_weights(m._W["net_token"], out=self.fold2)
        lm_w = m.lm_head.weight.data
        self.lm_fold.copy_(lm_w.float() * m._W["net_token"].norm.float()[None, :])

    # ---- the step bodies (run eagerly, or once under capture) ---------------------------------------------
    def _net_body(self):
        m = self.model
        spec = m._specs["net"]
        e = torch.empty((self.B, spec.D), dtype=m.dtype, device=m.device)
        ops.embed_sum_fwd(self.seq, m._W["net"].embed, e)  # the event sampled last
        engine.stack_decode(spec, m._W["net"], e, self.rope1, self.kv1, pos_dev=self.pos, folded=self.fold1, out=self.hidden)
        self.pos.add_(1)

    def _noise_body(self, generator=None):
        """the draws of one event: what torch.multinomial(probs_sort [B, V]) would draw at each of the T token positions"""
        for i in range(self.T):
            self.q_all[i].exponential_(1.0, generator=generator)

    def _tok_body(self, i: int, generator=None, draw: bool = True):
        m = self.model
        tspec, Wt = m._specs["net_token"], m._W["net_token"]
        self.kv2.len = i
        lm_w = m.lm_head.weight.data
        if self.lm_fold is not None:  # the stack's final RMSNorm rides on the lm_head projection
            if i == 0:
                h = engine.stack_decode(tspec, Wt, self.hidden, self.rope2, self.kv2, folded=self.fold2, final_norm=False)
            else:  # the embedding of the token just sampled is looked up inside the first projections
                h = engine.stack_decode(tspec, Wt, Wt.embed, self.rope2, self.kv2, folded=self.fold2, final_norm=False,
                                        x_ids=self.samples_in)
            ops.gemm_skinny(h, self.lm_fold, self.logits[:, : self.V], norm_eps=tspec.eps)
        else:
            x = self.hidden if i == 0 else Wt.embed.index_select(0, self.samples_in)
            h = engine.stack_decode(tspec, Wt, x, self.rope2, self.kv2)
            if ops.skinny_ok(h, tspec.D):
                ops.gemm_skinny(h, lm_w, self.logits[:, : self.V])
            else:
                ops.gemm_nt(h, lm_w, self.logits[:, : self.V])
        if i == 0 and not self.fused_sampler:
            self.seq.fill_(self.pad_id)
        if self.fused_sampler:
            # one launch instead of sample_top_p_k's ~25: the Exp(1) noise torch.multinomial would draw internally
            # (empty_like(probs).exponential_(1, generator)) is drawn here (eager form) or by the noise graph ahead of the
            # step (draw=False), the rest is mh_sample_top_p_k
            if draw:
                self.q_all[i].exponential_(1.0, generator=generator)
            ops.sample_top_p_k(self.logits, self.first_mask, self.lo_tab, self.hi_tab, self.ev, i, self.q_all[i], self.seq[:, i],
                               self.V, self.temp, self.top_p, self.top_k, out_b=self.samples_in,
                               out_c=self.ev if i == 0 else None, first_span=self.first_span, max_range=self.max_range[i],
                               ban_mask=self.ban, fill_rest=self.T - 1 if i == 0 else 0, fill_id=self.pad_id)
            return
        if i == 0:
            lo, hi = self.neg1, self.neg1
        else:
            lo, hi = self.lo_tab[self.ev, i].contiguous(), self.hi_tab[self.ev, i].contiguous()
        ops.masked_softmax(self.logits, lo, hi, self.first_mask, self.probs.view(self.B, self.V), self.V, self.temp)
        self.probs.mul_((self.ban == 0).view(1, 1, self.V))
        samples = m.sample_top_p_k(self.probs, self.top_p, self.top_k, generator=generator)  # (B, 1)
        self.seq[:, i] = samples[:, 0]
        if i == 0:
            self.ev.copy_(self.seq[:, 0])
        self.samples_in.copy_(self.seq[:, i])

    def _capture(self):
        with _CAPTURE_LOCK:
            self._capture_locked()

    def _capture_locked(self):
        # one eager pass on a side stream first: lazy initialisation (kernel attributes, allocator pools) must not
        # happen inside a capture
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._net_body()
            for i in range(self.T):
                self._tok_body(i, self.gen)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._pool = pool = torch.cuda.graph_pool_handle()
        self.g_net = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_net, pool=pool, capture_error_mode="thread_local"):
            self._net_body()
        if self.fused_sampler:
            # the event's draws in one graph (the only one that touches the generator) ...
            self.g_noise = torch.cuda.CUDAGraph()
            self.g_noise.register_generator_state(self.gen)  # philox seed/offset are read at replay time, offsets advance per replay
            # (its own memory pool: it replays on the noise stream WHILE the net graph runs on the caller's stream)
            with torch.cuda.graph(self.g_noise, capture_error_mode="thread_local"):
                self._noise_body(self.gen)
            # ... and all T token steps in another
            self.g_steps = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_steps, pool=pool, capture_error_mode="thread_local"):
                for i in range(self.T):
                    self._tok_body(i, draw=False)
            o0 = self.gen.get_offset()
            self.g_noise.replay()
            torch.cuda.synchronize()
            self._draw_inc = (self.gen.get_offset() - o0) // self.T
            assert self._draw_inc > 0 and self._draw_inc * self.T == self.gen.get_offset() - o0
            self.noise_stream = torch.cuda.Stream()
            self.noise_done = torch.cuda.Event()
            self.copy_stream = torch.cuda.Stream()
            self.steps_done = torch.cuda.Event()
            self.seq_host = torch.empty((self.B, self.T), dtype=torch.long).pin_memory()
        else:
            for i in range(self.T):
                self._capture_step(i)
        self.reset()

    def _capture_step(self, i: int):
        """a graph of token step i alone (the step-by-step form; with the fused sampler only tests / debugging use it: the
        noise then comes from draw_noise())"""
        g = torch.cuda.CUDAGraph()
        if not self.fused_sampler:
            g.register_generator_state(self.gen)
        with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
            self._tok_body(i, self.gen, draw=not self.fused_sampler)
        self.g_tok[i] = g

    # ---- driver interface ---------------------------------------------------------------------------------
    def reset(self):
        self.kv1.len = 0
        self.kv2.len = 0
        self.pos.zero_()

    def prefill(self, tokens: torch.Tensor) -> None:
        """tokens [B, S, T]: causal forward over the prompt from an empty cache; leaves hidden = last position."""
        m = self.model
        spec = m._specs["net"]
        B, S, T = tokens.shape
        e = torch.empty((B * S, spec.D), dtype=m.dtype, device=m.device)
        ops.embed_sum_fwd(tokens.contiguous().view(B * S, T), m._W["net"].embed, e)
        self.kv1.len = 0
        y = engine.stack_prefill(spec, m._W["net"], e, B, S, self.rope1, self.kv1, folded=self.fold1)
        self.hidden.copy_(y.view(B, S, spec.D)[:, -1])
        self.pos.fill_(S)

    def begin(self, generator) -> None:
        """take over the caller's random stream (None = the device's default generator)"""
        self._user_gen = generator
        self._noise_pending = False
        self._noise_read = False
        if self.gen is not None:
            src = generator if generator is not None else torch.cuda.default_generators[self.model.device.index or 0]
            with _CAPTURE_LOCK:
                self.gen.set_state(src.get_state())
                self._off = self.gen.get_offset()

    def end(self) -> None:
        """hand the advanced random stream back to the caller's generator"""
        if self.g_noise is not None:
            if self._noise_pending:  # draws made ahead for an event that was never sampled: not consumed
                self.noise_stream.synchronize()
                self._noise_pending = False
            with _CAPTURE_LOCK:
                self.gen.set_offset(self._off)
        if self.gen is not None:
            dst = self._user_gen if self._user_gen is not None else torch.cuda.default_generators[self.model.device.index or 0]
            with _CAPTURE_LOCK:
                dst.set_state(self.gen.get_state())
        self._user_gen = None

    def net_step(self) -> None:
        """decode the event in ``seq`` (the one just sampled) at the next position; hidden <- net output"""
        if self.g_net is not None:
            self.g_net.replay()
        else:
            self._net_body()
        self.kv1.len += 1

    def tok_step(self, i: int) -> None:
        """sample token position i of the current event into seq[:, i] (and ev for i == 0) -- the step-by-step form.
        With the fused sampler the step READS the event's Exp(1) variates instead of drawing them: they come from
        draw_noise() (called here for position 0 when the caller has not done so -- without it the step would sample against
        stale variates and leave the generator where it was), and the caller reports the draws the reference loop would have
        made with consumed(n_steps) once the event is complete (sample_event() does both)."""
        if self.use_graphs:
            if self.g_tok[i] is None:
                with _CAPTURE_LOCK:
                    self._capture_step(i)
            if self.g_noise is not None and i == 0:
                if self._noise_pending and self._noise_read:
                    # the pending variates were already read by an earlier event's steps and never reported with consumed():
                    # sampling against them again would repeat that event's draws and leave the generator where it was
                    raise RuntimeError("DecodeSession.tok_step(0): the previous event's draws were not reported -- call "
                                       "consumed(n_steps) after the last tok_step of an event (sample_event() does both)")
                if not self._noise_pending:
                    self.draw_noise()
            if self.g_noise is not None:
                self._noise_read = True
            if self.g_noise is not None and self._noise_pending:
                torch.cuda.current_stream().wait_event(self.noise_done)
            if self.fused_sampler:
                self.g_tok[i].replay()
            else:  # (the step graph of the unfused sampler carries the generator state)
                with _CAPTURE_LOCK:
                    self.g_tok[i].replay()
            self._steps_recorded = False
        else:
            self._tok_body(i, self._user_gen)

    # ---- one event per call: the form generate() drives ----------------------------------------------------
    def draw_noise(self) -> None:
        """(fused graphs) draw the next event's T x [B, V] variates on the noise stream, starting at the offset the reference
        loop would be at; they overlap whatever the caller queues next on its own stream (the net step)"""
        if self.g_noise is None:
            return
        # the token steps that read the previous draws must be done first: the event recorded right behind the steps graph
        # when there is one (NOT the whole stream: the net step queued after it is what the draws should overlap)
        if _NOISE_INLINE:
            pass
        elif self._steps_recorded:
            self.noise_stream.wait_event(self.steps_done)
        else:
            self.noise_stream.wait_stream(torch.cuda.current_stream())
        self._noise_read = False
        if _NOISE_INLINE:  # (A/B: the draws on the caller's stream, no cross-stream dependency)
            with _CAPTURE_LOCK:  # (generator-state calls assert "not capturing" as well)
                self.gen.set_offset(self._off)
                self.g_noise.replay()
            self.noise_done.record(torch.cuda.current_stream())
        else:
            with torch.cuda.stream(self.noise_stream):
                with _CAPTURE_LOCK:
                    self.gen.set_offset(self._off)
                    self.g_noise.replay()
                self.noise_done.record(self.noise_stream)
        self._noise_pending = True

    def consumed(self, n_steps: int) -> None:
        """the reference loop ran n_steps sampling calls for the event just sampled: its generator stands n_steps draws on"""
        if self.g_noise is not None:
            self._off += n_steps * self._draw_inc
            self._noise_pending = False
            self._noise_read = False

    def n_steps_of(self, ids) -> tuple:
        """(number of sampling calls the reference makes for an event whose first tokens are `ids`, all rows ended?) --
        the break rule of midi_model.py:232-235: the inner loop stops after position i iff every live row's event has
        exactly i parameters (vacuously true at i == 1 when no row is live)"""
        tok = self.model.tokenizer
        arity = self.model._grammar()[3]
        alive = [arity[t] for t in ids if t != tok.eos_id]
        if not alive:
            return 2, True
        return (alive[0] + 1 if all(a == alive[0] for a in alive) else self.T), False

    def sample_event(self, then_net: bool = False):
        """sample the T tokens of the next event from `hidden`; -> (np.ndarray [B, T] int64, all rows ended?).
        ``then_net`` (graph form): the net step over the sampled event is queued right behind the token steps, BEFORE the host
        waits for the tokens -- it depends on nothing the host decides -- so the device never idles through the host's
        round trip; the tokens come back through a copy stream that waits for the token steps only.  (If every row turns out
        to have ended, that net step was wasted work on a session about to be reset.)"""
        if self.g_steps is not None:
            if not self._noise_pending:
                self.draw_noise()
            cur = torch.cuda.current_stream()
            cur.wait_event(self.noise_done)
            self.g_steps.replay()
            self.steps_done.record(cur)
            self._steps_recorded = True
            if then_net:
                self.net_step()
                self.copy_stream.wait_event(self.steps_done)
                with torch.cuda.stream(self.copy_stream):
                    if _COPY_PAGEABLE:  # (A/B: a blocking pageable copy on the copy stream instead of pinned + stream sync)
                        event = self.seq.cpu().numpy().copy()
                    else:
                        self.seq_host.copy_(self.seq, non_blocking=True)
                if not _COPY_PAGEABLE:
                    self.copy_stream.synchronize()  # the one host sync per event
                    event = self.seq_host.numpy().copy()
            else:
                event = self.seq.cpu().numpy().copy()  # the one host sync per event
            n, end_all = self.n_steps_of(event[:, 0].tolist())
            self.consumed(n)
            return event, end_all
        n_steps, end_all, i = self.T, False, 0
        while i < n_steps:
            self.tok_step(i)  # ... lm_head -> masked softmax -> sample_top_p_k -> seq[:, i]
            if i == 0:
                n_steps, end_all = self.n_steps_of(self.ev.tolist())  # (a host sync)
            i += 1
        return self.seq.cpu().numpy().copy(), end_all  # (a CPU tensor would share memory with the session buffer)
