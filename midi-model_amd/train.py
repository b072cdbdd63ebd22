"""The data-parallel training step of the reference (train.py:106-206, Trainer wiring :461-474) on HIP.

``TrainMIDIModel`` keeps the reference's names (``training_step``, ``validation_step``,
``configure_optimizers``, ``compute_accuracy``) but, with no Lightning underneath, the step is explicit:

    loss = model.training_step(batch)      # forward + backward; gradients land in one flat buffer
    model.optimizer_step()                 # every `accumulate_grad_batches` micro-batches:
                                           #   [all-reduce done] -> global-norm clip(1.0) -> fused AdamW -> LR schedule

Forward/backward are the explicit schedules of ``engine.py``; lm_head + cross-entropy run chunked so the
(B*S*8, vocab) logits tensor (1.8 GB in bf16 at B=16, S=2048) is never materialised.  Data parallelism is one
process per GPU: gradients are averaged with bucketed asynchronous all-reduces (torch.distributed "nccl" =
RCCL over xGMI; "gloo" in the CPU tests) issued on a side stream as backward finishes each contiguous range of
the flat gradient buffer, so communication overlaps the rest of backward (reference: DDP inside Lightning).
"""
from __future__ import annotations

import math
import os
import random
from typing import Callable, List, Optional, Tuple

import torch

from . import engine, ops
from .config import MIDIModelConfig
from .model import MIDIModel


def lr_lambda(step: int, warmup: float, max_step: float) -> float:
    """get_linear_schedule_with_warmup (train.py:93-103)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(max_step - step) / float(max(1, max_step - warmup)))


class GradReducer:
    """Bucketed, overlapped gradient averaging over a flat buffer.

    ``ready(lo, hi)`` announces that flat[lo:hi] holds final gradients; announcements arrive back to front
    (backward order).  Adjacent ranges are merged until ``bucket_bytes`` is reached, then one all-reduce(SUM)
    of the merged range is launched asynchronously on the communication stream; ``finish()`` flushes the tail,
    waits for everything and leaves gradients divided by world size.  Device-agnostic (CPU tensors + gloo in
    the tests, HIP tensors + RCCL on the GPU)."""

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_bytes: int = 32 << 20, comm=None, force: bool = False):
        """``comm``: an ``MHComm`` (comm.py) -- the buckets then go out through the library's own RCCL communicator
        (``mh_comm_allreduce``: the mean as ONE pre-multiplied-sum collective on the communication stream) instead of
        ``torch.distributed``.  ``force``: run the whole bucketed path even when there is one rank (the real backend's call
        sequence on a one-GPU box: tests, bench.py's contention probe)."""
        import torch.distributed as dist
        self.dist = dist
        self.flat = flat_grad
        self.group = group
        self.comm = comm
        self.force = bool(force)
        if comm is not None:
            if not flat_grad.is_cuda:
                raise ValueError("GradReducer: an MHComm (RCCL) communicator needs the flat gradient buffer on the device")
            self.world = comm.world
        else:
            self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
            if self.force and not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("GradReducer(force=True) runs the real collective at world size 1: it needs `comm` or an "
                                   "initialised torch.distributed process group")
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self.pending: Optional[Tuple[int, int]] = None
        self.works: list = []
        self.launched: List[Tuple[int, int]] = []
        self.comm_stream = torch.cuda.Stream() if flat_grad.is_cuda else None
        # bench.py: with `profile` set, finish() brackets its wait for the communication stream with HIP events on the
        # compute stream -- the time between them is the all-reduce time NOT hidden behind backward -- and counts the bytes
        self.profile = False
        self.stats: list = []          # (start event | None, end event | None, bytes, launches) per optimiser step
        self._bytes = 0
        # the ranges one accumulation window must cover (None = all of `flat`); a backward that legitimately skips ranges
        # (frozen parameters) sets it to the trainable ranges
        self.expected: Optional[List[Tuple[int, int]]] = None

    def ready(self, lo: int, hi: int):
        if self.world == 1 and not self.force:
            return
        if self.pending is None:
            self.pending = (lo, hi)
        else:
            plo, phi = self.pending
            if hi == plo:
                self.pending = (lo, phi)
            elif lo == phi:
                self.pending = (plo, hi)
            else:  # not adjacent: ship what we have
                self._launch(plo, phi)
                self.pending = (lo, hi)
        plo, phi = self.pending
        if phi - plo >= self.bucket_elems:
            self._launch(plo, phi)
            self.pending = None

    def _launch(self, lo: int, hi: int):
        buf = self.flat[lo:hi]
        self.launched.append((lo, hi))
        self._bytes += buf.numel() * buf.element_size()
        if self.comm is not None:
            # library-owned communicator: ordered on the communication stream, finish() waits for that stream
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            self.comm.allreduce_(buf, mean=True, stream=self.comm_stream)
        elif self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                buf.div_(self.world)
                self.works.append(self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            buf.div_(self.world)
            self.works.append(self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        if self.world == 1 and not self.force:
            return
        # every element of the expected ranges (default: the whole flat gradient buffer -- lm_head, token-level layers and
        # embedding, event-level layers and embedding, norm vectors) goes out exactly once per accumulation window.  Checked
        # BEFORE the tail is launched; a failure still drains what is in flight and clears the window's state, so one bad
        # window does not poison the following ones.
        cover = sorted(self.launched + ([self.pending] if self.pending is not None else []))
        problem = None
        if cover:
            def merge(ranges):  # adjacent ranges fused: [(0, 40), (40, 100)] and [(0, 100)] are the same cover
                out = []
                for a in sorted(ranges):
                    if out and out[-1][1] == a[0]:
                        out[-1] = (out[-1][0], a[1])
                    else:
                        out.append(tuple(a))
                return out
            want = merge(self.expected if self.expected is not None else [(0, self.flat.numel())])
            merged = merge(cover)
            overlap = any(a[1] > b[0] for a, b in zip(cover, cover[1:]))
            if overlap or merged != want:
                problem = (f"GradReducer: the announced ranges do not tile the expected ranges {want[:4]}: "
                           f"merged {merged[:6]}, overlap={overlap}")
        if problem is not None:
            self.pending = None
            self._drain()
            raise RuntimeError(problem)
        if self.pending is not None:
            self._launch(*self.pending)
            self.pending = None
        ev = None
        if self.profile and self.comm_stream is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self.works:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.profile:
            if ev is not None:
                ev[1].record()
            self.stats.append((ev[0] if ev else None, ev[1] if ev else None, self._bytes, len(self.launched)))
        self._bytes = 0
        self.works.clear()
        self.launched.clear()

    def _drain(self):
        """wait for whatever is in flight and forget the window (error path of finish())"""
        for w in self.works:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._bytes = 0
        self.works.clear()
        self.launched.clear()


class TrainMIDIModel(MIDIModel):
    def __init__(self, config: MIDIModelConfig, lr=2e-4, weight_decay=0.01, warmup=1e3, max_step=1e6,
                 sample_seq=False, gen_example_interval=1, example_batch=8, accumulate_grad_batches=2,
                 gradient_clip_val=1.0, ce_chunk_rows=32768, bucket_mb=32):
        super().__init__(config)
        self.lr, self.weight_decay, self.warmup, self.max_step = lr, weight_decay, warmup, max_step
        self.sample_seq = sample_seq
        self.gen_example_interval, self.example_batch = gen_example_interval, example_batch
        self.accumulate_grad_batches = accumulate_grad_batches
        self.gradient_clip_val = gradient_clip_val
        self.ce_chunk_rows = ce_chunk_rows
        self.bucket_mb = bucket_mb
        self.betas, self.eps = (0.9, 0.99), 1e-8
        self.global_step = 0       # optimiser steps taken
        self._micro = 0            # micro-batches since the last optimiser step
        self._opt = None
        self._reducer = None
        self.last_grad_norm = None
        self.process_group = None
        self.comm = None           # MHComm (comm.py): the gradient exchange through the library's own RCCL communicator
        # memory for time: the forward does not keep the SwiGLU activations, the backward recomputes them from gate|up (identical
        # bits, one extra elementwise pass per layer: ~1 % of a step).  For shapes whose activations crowd the 288 GB -- the
        # 2x-hidden large shape at 16 x 4096 per GPU peaks at 306 of 309 GB without it, and RCCL needs room for its channel buffers
        self.lean_activations = False
        # r06: the training forward / backward with the RMSNorms folded around the projections (engine.layer_forward_train_folded):
        # no passes over the residual stream for the norms' forward, no normalised activations kept, the norm weights' gradients
        # out of the weight-gradient reductions.  bf16 only (engine.train_fold_ok); False = the r01-r05 schedule.
        self.fold_train_norms = True
        self.force_reduce = False  # run the bucketed exchange even with one rank (tests, bench.py's contention probe)
        self._lora = None          # LoraAdapter while fine-tuning adapters on a frozen base (add_adapter)

    # ----------------------------------------------------------------------------------- optimiser
    def configure_optimizers(self):
        """AdamW(lr, betas=(0.9, 0.99), eps=1e-8); weight decay on every parameter whose NAME contains neither
        'bias' nor 'norm' (train.py:121-151) — in the flat layout that is exactly the matrix region."""
        self._require_gpu()
        flat = self._flat
        self._opt = {
            "m": torch.zeros_like(flat), "v": torch.zeros_like(flat),
            "sumsq": torch.zeros(1, dtype=torch.float32, device=flat.device),
            "partial": torch.empty(1024, dtype=torch.float32, device=flat.device),
            "coef": torch.ones(1, dtype=torch.float32, device=flat.device),
            "norm": torch.zeros(1, dtype=torch.float32, device=flat.device),
        }
        for name, (off, n, grp) in self._offsets.items():
            assert (grp == "norm") == any(nd in name for nd in ("bias", "norm")), name
        return self._opt

    def current_lr(self) -> float:
        return self.lr * lr_lambda(self.global_step, self.warmup, self.max_step)

    def optimizer_step(self):
        """clip_grad_norm_(1.0) -> AdamW -> scheduler.step(), all on device (no host sync)."""
        if self._lora is not None:
            return self._lora_optimizer_step()
        if self._opt is None:
            self.configure_optimizers()
        if self._reducer is not None:
            self._reducer.finish()
        o = self._opt
        g = self.grad_buffer()
        if self.gradient_clip_val is not None and self.gradient_clip_val > 0:
            ops.sumsq(g, o["partial"], o["sumsq"], False)
            ops.clip_coef(o["sumsq"], float(self.gradient_clip_val), o["coef"], o["norm"])
            self.last_grad_norm = o["norm"]
            coef = o["coef"]
        else:
            coef = None
        lr = self.current_lr()
        step = self.global_step + 1
        bc1, bc2 = 1.0 - self.betas[0] ** step, 1.0 - self.betas[1] ** step
        nm = self._n_mat
        flat = self._flat
        ops.adamw(flat[:nm], g[:nm], o["m"][:nm], o["v"][:nm], lr, self.betas[0], self.betas[1], self.eps,
                  self.weight_decay, bc1, bc2, coef)
        ops.adamw(flat[nm:], g[nm:], o["m"][nm:], o["v"][nm:], lr, self.betas[0], self.betas[1], self.eps, 0.0,
                  bc1, bc2, coef)
        self.weights_written()  # (raw-pointer write: data derived from the weights -- MIDIModel.folded_weights -- is stale now)
        self.global_step += 1
        self._micro = 0

    # ------------------------------------------------------------------------ training-state resume
    _NO_DECAY = ("bias", "norm")  # train.py:123

    def _optimizer_param_order(self, names):
        """the parameter order of the reference's optimizer (train.py:121-131): named_parameters order, the decayed group
        first, then the names containing 'bias' or 'norm'"""
        names = list(names)
        return [n for n in names if not any(nd in n for nd in self._NO_DECAY)] + \
               [n for n in names if any(nd in n for nd in self._NO_DECAY)]

    def training_state(self) -> dict:
        """What ``trainer.fit(..., ckpt_path=opt.resume)`` (train.py:475-479) needs to continue a run, in the layout of a
        Lightning ``.ckpt``: ``state_dict``, ``global_step``, ``optimizer_states[0]`` = the ``torch.optim.AdamW.state_dict()`` of
        the reference's two parameter groups (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), ``lr_schedulers[0]`` = the
        LambdaLR position (with the ``lr_lambdas`` / ``verbose`` / ``_get_lr_called_within_step`` keys
        ``LambdaLR.load_state_dict`` expects) and the ``pytorch-lightning_version`` / ``callbacks`` keys (no ``loops``: see below).  Tested direction: a reference-layout checkpoint -> ``load_training_state`` and our own round trip, and
        the scheduler / optimizer dictionaries loaded into the real torch objects (tests/test_host_logic.py); Lightning itself is
        not installable here.  Ours on top: the accumulation phase (``mh_micro``) and, inside an accumulation window, the gradient
        accumulated so far."""
        if self._lora is not None:
            raise RuntimeError("training_state: adapter training keeps its state in the adapter (save_adapter)")
        if self._opt is None:
            self.configure_optimizers()
        names = [n for n, _ in self.named_parameters()]
        order = self._optimizer_param_order(names)
        n_decay = sum(1 for n in names if not any(nd in n for nd in self._NO_DECAY))
        m, v = self._opt["m"], self._opt["v"]
        state = {}
        for i, n in enumerate(order):
            off, cnt, _ = self._offsets[n]
            state[i] = {"step": torch.tensor(float(self.global_step)),
                        "exp_avg": m[off:off + cnt].detach().cpu().clone(), "exp_avg_sq": v[off:off + cnt].detach().cpu().clone()}
        shapes = {n: p.shape for n, p in self.named_parameters()}
        for i, n in enumerate(order):
            state[i]["exp_avg"] = state[i]["exp_avg"].view(shapes[n])
            state[i]["exp_avg_sq"] = state[i]["exp_avg_sq"].view(shapes[n])
        lr_now = self.current_lr()
        group = dict(lr=lr_now, initial_lr=self.lr, betas=self.betas, eps=self.eps, amsgrad=False, maximize=False, foreach=None,
                     capturable=False, differentiable=False, fused=None)
        out = {
            "state_dict": {k: t.detach().cpu().clone() for k, t in self.state_dict().items()},
            "global_step": int(self.global_step), "epoch": 0,
            "optimizer_states": [{"state": state, "param_groups": [
                dict(group, weight_decay=self.weight_decay, params=list(range(n_decay))),
                dict(group, weight_decay=0.0, params=list(range(n_decay, len(order))))]}],
            # torch.optim.lr_scheduler.LambdaLR.state_dict(): the lambdas themselves are not pickled (plain functions -> None)
            # and load_state_dict pops "lr_lambdas", so the key must exist for the reference's trainer to resume from this file
            "lr_schedulers": [{"last_epoch": int(self.global_step), "_step_count": int(self.global_step) + 1,
                               "base_lrs": [self.lr, self.lr], "_last_lr": [lr_now, lr_now], "lr_lambdas": [None, None],
                               "verbose": False, "_get_lr_called_within_step": False}],
            "pytorch-lightning_version": "2.4.0",
            # (no "loops" key: Lightning's restore_loops skips a checkpoint that has none, while an EMPTY dict would be indexed
            #  with ["fit_loop"]; Lightning's own global_step then restarts at 0 -- ours and the scheduler's come from the keys above)
            "callbacks": {},
            "hyper_parameters": dict(lr=self.lr, weight_decay=self.weight_decay, warmup=self.warmup, max_step=self.max_step),
            "mh_micro": int(self._micro),
        }
        if self._micro > 0 and self._flat_grad is not None:
            out["mh_grad"] = self._flat_grad.detach().cpu().clone()
        return out

    def save_training_state(self, path: str) -> None:
        torch.save(self.training_state(), path)

    def load_training_state(self, state, trust_checkpoint: bool = False) -> "TrainMIDIModel":
        """Resume from ``training_state()`` or from a Lightning ``.ckpt`` payload of the reference's trainer (a path or the
        loaded dict): weights, AdamW moments (mapped back to parameters through the reference's group order), the step count
        the bias corrections and the LR schedule run on, the accumulation phase.  Strict: a moment tensor that is missing or has
        the wrong shape raises."""
        if isinstance(state, (str, os.PathLike)):
            import pickle
            try:
                state = torch.load(state, map_location="cpu", weights_only=True)
            except pickle.UnpicklingError as e:
                # a Lightning .ckpt with callback / hyper-parameter objects outside torch's allow-list: same per-call trust as
                # MIDIModel.from_checkpoint (the reference's trainer unpickles whatever the file names)
                if not trust_checkpoint:
                    raise RuntimeError(f"{state} holds pickled objects beyond tensors ({e}); pass trust_checkpoint=True to unpickle "
                                       "it as Lightning does (only for a file you trust)") from e
                import warnings
                warnings.warn(f"load_training_state: unpickling {state} with weights_only=False (trust_checkpoint=True)", stacklevel=2)
                state = torch.load(state, map_location="cpu", weights_only=False)
        if "optimizer_states" not in state or "state_dict" not in state:
            raise RuntimeError("load_training_state: not a training checkpoint (needs state_dict + optimizer_states)")
        self.load_checkpoint_state(state)
        if self._opt is None:
            self.configure_optimizers()
        # the checkpoint's own key order = the saving module's named_parameters order (state_dict of parameters only)
        own = dict(self.named_parameters())
        keys = list(state["state_dict"].keys())
        prefixes = ("model.", "base_model.model.", "module.", "_orig_mod.")
        for _ in range(8):
            if set(keys) & set(own):
                break
            hit = next((p for p in prefixes if keys and all(k.startswith(p) for k in keys)), None)
            if hit is None:
                break
            keys = [k[len(hit):] for k in keys]
        names = [k for k in keys if k in own]
        order = self._optimizer_param_order(names)
        opt = state["optimizer_states"][0]
        groups = opt["param_groups"]
        idx = [i for g in groups for i in g["params"]]
        if len(idx) != len(order) or set(order) != set(own):
            raise RuntimeError(f"load_training_state: the optimizer holds {len(idx)} parameters, the model {len(own)} "
                               f"({len(order)} of them in the checkpoint)")
        m, v = self._opt["m"], self._opt["v"]
        steps = set()
        for i, n in zip(idx, order):
            st = opt["state"].get(i)
            if st is None:  # torch creates a parameter's state at its first step with a gradient
                if int(state.get("global_step", 0)) != 0:
                    raise RuntimeError(f"load_training_state: no optimizer state for {n} in a checkpoint at step {state.get('global_step')}")
                continue
            off, cnt, _ = self._offsets[n]
            for key, dst in (("exp_avg", m), ("exp_avg_sq", v)):
                t = st[key]
                if tuple(t.shape) != tuple(own[n].shape):
                    raise RuntimeError(f"load_training_state: {key} of {n} has shape {tuple(t.shape)}, expected {tuple(own[n].shape)}")
                dst[off:off + cnt].copy_(t.reshape(-1).to(device=dst.device, dtype=dst.dtype))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise RuntimeError(f"load_training_state: parameters disagree on the step count {sorted(steps)} (the fused AdamW keeps one)")
        self.global_step = int(state.get("global_step", steps.pop() if steps else 0))
        self._micro = int(state.get("mh_micro", 0))
        if "mh_grad" in state:
            self.grad_buffer().copy_(state["mh_grad"].to(self._flat.device, self._flat.dtype))
        elif self._micro > 0:
            raise RuntimeError("load_training_state: the checkpoint was taken inside an accumulation window but holds no gradient")
        # the schedule continues from the checkpoint's base learning rate (LambdaLR.state_dict()["base_lrs"]: the lambdas
        # themselves -- warm-up length, last step -- are not pickled by torch and stay as this object was constructed)
        sched = (state.get("lr_schedulers") or [{}])[0] or {}
        if sched.get("base_lrs"):
            self.lr = float(sched["base_lrs"][0])
        decay = [g.get("weight_decay") for g in groups if g.get("weight_decay")]
        if decay:
            self.weight_decay = float(decay[0])
        hp = state.get("hyper_parameters") or {}  # (ours: training_state() writes the schedule's shape too)
        for k in ("warmup", "max_step"):
            if k in hp:
                setattr(self, k, hp[k])
        return self

    # ------------------------------------------------------------------------------------- LoRA
    def add_adapter(self, lora_config=None, **kwargs):
        """train.py:439-449: ``model.requires_grad_(False); model.add_adapter(LoraConfig(r=64, lora_alpha=128,
        target_modules=[q,o,k,v,gate,up,down], lora_dropout=0))``.  Accepts a config object with those attributes (peft's
        LoraConfig), a dict, or keyword arguments.  From here on training_step / optimizer_step update the adapters only;
        the base weights stay as loaded (lora.py)."""
        from .lora import DEFAULT_TARGETS, LoraAdapter
        self._require_gpu()
        cfg = dict(r=64, lora_alpha=128, target_modules=DEFAULT_TARGETS, lora_dropout=0.0)
        src = lora_config if isinstance(lora_config, dict) else (
            {k: getattr(lora_config, k) for k in cfg if hasattr(lora_config, k)} if lora_config is not None else {})
        cfg.update({k: v for k, v in src.items() if k in cfg and v is not None})
        cfg.update({k: v for k, v in kwargs.items() if k in cfg})
        if self._lora is not None:
            raise RuntimeError("an adapter is already attached (merge_and_unload() first)")
        self.requires_grad_(False)
        self._lora = LoraAdapter(self, cfg["r"], cfg["lora_alpha"], tuple(cfg["target_modules"]), cfg["lora_dropout"],
                                 kwargs.get("generator"))
        self._micro = 0
        return self._lora

    def save_adapter(self, directory: str) -> None:
        """adapter_config.json + adapter_model.safetensors in peft's layout (readable by MIDIModel.load_merge_lora)"""
        if self._lora is None:
            raise RuntimeError("no adapter attached")
        self._lora.save(directory)

    def save_peft(self, save_dir: str) -> None:
        """the reference's name for save_adapter (train.py:234-244)"""
        self.save_adapter(save_dir)

    def gen_example(self, save_dir: str, prompt=None, max_len: int = 512, generator=None):
        """train.py:208-232 without the Lightning / dataset globals: `example_batch` sequences generated from BOS and, when
        `prompt` (int array [T, 8]) is given, as many continuations of its first 256 events, written under
        save_dir/sample/<global_step>/ as <k>_<i>.npy token arrays -- and as .mid / .png next to them when the tokenizer
        carries the reference's codec (detokenize / midi2img) and the reference's `MIDI` module is importable (CPU format
        code, out of this build's scope: used as it is).  Returns the list of generated arrays."""
        import numpy as np
        base_dir = os.path.join(save_dir, "sample", str(self.global_step))
        os.makedirs(base_dir, exist_ok=True)
        try:
            import MIDI  # the reference's module, when on sys.path
        except Exception:
            MIDI = None
        has_codec = getattr(type(self.tokenizer), "detokenize", None) is not None and \
            getattr(self.tokenizer, "_no_codec", None) is None
        out = []
        cases = [(0, None)] if prompt is None else [(0, None), (1, np.asarray(prompt)[:256].astype(np.int64))]
        for k, pr in cases:
            seqs = self.generate(pr, batch_size=self.example_batch, max_len=max_len, generator=generator)
            for i, seq in enumerate(seqs):
                seq = np.asarray(seq)
                np.save(os.path.join(base_dir, f"{k}_{i}.npy"), seq)
                out.append(seq)
                if has_codec:
                    score = self.tokenizer.detokenize(seq)
                    self.tokenizer.midi2img(score).save(os.path.join(base_dir, f"{k}_{i}.png"))
                    if MIDI is not None:
                        with open(os.path.join(base_dir, f"{k}_{i}.mid"), "wb") as f:
                            f.write(MIDI.score2midi(score))
        return out

    def merge_and_unload(self):
        """fold the adapter into the weights (W <- W + scale * B @ A) and drop it; returns self"""
        if self._lora is not None:
            self._lora.materialize(self)
            self._lora = None
            self.requires_grad_(True)
        return self

    def _lora_optimizer_step(self):
        import torch.distributed as dist
        lo = self._lora
        lo.compute_grads(self)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            lo.grad.div_(dist.get_world_size(self.process_group))
            dist.all_reduce(lo.grad, group=self.process_group)
        o = lo.optimizer_state()
        if self.gradient_clip_val is not None and self.gradient_clip_val > 0:
            ops.sumsq(lo.grad, o["partial"], o["sumsq"], False)
            ops.clip_coef(o["sumsq"], float(self.gradient_clip_val), o["coef"], o["norm"])
            self.last_grad_norm = o["norm"]
            coef = o["coef"]
        else:
            coef = None
        lr = self.current_lr()
        step = self.global_step + 1
        bc1, bc2 = 1.0 - self.betas[0] ** step, 1.0 - self.betas[1] ** step
        # (names lora_A / lora_B contain neither "bias" nor "norm": weight decay applies, train.py:121-151)
        ops.adamw(lo.flat, lo.grad, o["m"], o["v"], lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                  bc1, bc2, coef)
        lo.materialize(self)  # live weights follow the adapters at once: generate()/forward()/state_dict() right after a step
        self.weights_written()
        self.global_step += 1
        self._micro = 0

    def _train_fold(self, spec, rows: torch.Tensor, backward: bool):
        """the folded weights stack_forward should run with: this step's (re-derived after every update) when the training fold
        applies, the kept inference fold for a forward-only pass over enough rows, else None"""
        if backward:
            if self.fold_train_norms and self._lora is None and engine.train_fold_ok(spec, rows):
                return self.folded_weights(spec.name)
            return None
        return self._folded_for(spec, rows) if spec.kind == "event" else None

    def zero_grad(self, set_to_none: bool = False):
        if self._flat_grad is not None:
            self._flat_grad.zero_()
        self._micro = 0

    # --------------------------------------------------------------------------------- fused step
    def _loss_and_backward(self, batch: torch.Tensor, backward: bool, want_acc: bool = False):
        """train.py:168-188 (+ its backward).  batch (B, S+1, 8) int64.  Returns (loss[1] fp32 device tensor, acc)."""
        self._require_gpu()
        if self._lora is not None and self._lora.dirty:
            self._lora.materialize(self)  # live weights <- base + scale * B @ A
        tok = self.tokenizer
        dev, dty = self.device, self.dtype
        self._check_ids(batch)
        batch = batch.to(device=dev, dtype=torch.long)
        x = batch[:, :-1].contiguous()
        y = batch[:, 1:].contiguous()
        B, S, T = x.shape
        spec, tspec = self._specs["net"], self._specs["net_token"]
        Wn, Wt = self._W["net"], self._W["net_token"]
        D = spec.D
        V, Vp = tok.vocab_size, self.vocab_padded
        M = B * S

        # ---- forward: event-level net
        e = torch.empty((M, D), dtype=dty, device=dev)
        ops.embed_sum_fwd(x.view(M, T), Wn.embed, e)
        hidden, ctx_net = engine.stack_forward(spec, Wn, e, B, S, self.rope("net"), save=backward, lean=self.lean_activations,
                                               folded=self._train_fold(spec, e, backward))
        del e
        sel = None
        if self.sample_seq:  # train.py:172-175: keep the last position + up to 127 random others
            idx = [-1] + random.sample(list(range(S - 2)), min(127, (S - 2) // 2))
            sel = torch.tensor([i % S for i in idx], dtype=torch.long, device=dev)
            hidden_t = hidden.view(B, S, D)[:, sel].reshape(-1, D).contiguous()
            y_t = y[:, sel].reshape(-1, T).contiguous()
        else:
            hidden_t = hidden
            y_t = y.view(M, T)
        N = hidden_t.shape[0]
        R = N * T

        # ---- forward: token-level net over [hidden ; embed(y[:, :7])]
        seq = torch.empty((N, T, D), dtype=dty, device=dev)
        ops.concat_tok_fwd(hidden_t, y_t, Wt.embed, seq, T)
        h, ctx_tok = engine.stack_forward(tspec, Wt, seq.view(R, D), N, T, self.rope("net_token"), save=backward,
                                          lean=self.lean_activations, folded=self._train_fold(tspec, seq.view(R, D), backward))
        del seq

        # ---- lm_head + cross-entropy (+ their backward), chunked over rows
        targets = y_t.reshape(R)
        cnt = torch.empty(1, dtype=torch.float32, device=dev)
        inv = torch.empty(1, dtype=torch.float32, device=dev)
        ops.count_valid(targets, tok.pad_id, cnt, inv)
        # Lightning's automatic optimisation divides every micro-batch loss by accumulate_grad_batches before its
        # backward (loops/optimization/automatic.py, ClosureResult normalize=...), so the window's gradient is the MEAN of
        # the micro-batch gradients and gradient_clip_val acts on that mean; the reported loss stays per micro-batch.
        nacc = max(1, int(self.accumulate_grad_batches))
        bwd_scale = inv / nacc if (backward and nacc > 1) else inv
        row_loss = torch.empty(R, dtype=torch.float32, device=dev)
        argmax = torch.empty(R, dtype=torch.long, device=dev) if want_acc else None
        lm_w = self.lm_head.weight.data
        accumulate = backward and self._micro > 0
        if backward:
            self.grad_buffer()
            dh = torch.empty((R, D), dtype=dty, device=dev)
        chunk = max(T, (self.ce_chunk_rows // T) * T)
        logits = torch.empty((min(chunk, R), Vp), dtype=dty, device=dev)
        first = True
        for r0 in range(0, R, chunk):
            r1 = min(R, r0 + chunk)
            lg = logits[: r1 - r0]
            ops.gemm_nt(h[r0:r1], lm_w, lg[:, :V])
            ops.cross_entropy(lg, V, targets[r0:r1], row_loss[r0:r1], lg if backward else None, bwd_scale,
                              argmax[r0:r1] if want_acc else None, tok.pad_id)
            if backward:
                # d h = dlogits @ W_lm and d W_lm += dlogits^T @ h, operands read as they lie (padding columns of
                # dlogits are zero, so the contraction may run to the 8-aligned V)
                ops.gemm_nt(lg, lm_w, dh[r0:r1], K=V, tb=True)
                ops.gemm_nt(lg, h[r0:r1], self._g_lm, K=r1 - r0, ta=True, tb=True,
                            beta=0.0 if (first and not accumulate) else 1.0)
                first = False
        loss_sum = torch.empty(1, dtype=torch.float32, device=dev)
        ops.sum_f32(row_loss, loss_sum)
        loss = loss_sum * inv
        acc = None
        if want_acc:
            keep = targets != tok.pad_id
            acc = ((argmax == targets) & keep).sum().float() / keep.sum()
        if not backward:
            return loss, acc

        # ---- backward, announcing finished gradient ranges to the reducer back to front
        red = self._reducer_for_step()
        off_lm, n_lm, _ = self._offsets["lm_head.weight"]
        self._announce(red, off_lm, off_lm + n_lm)

        def tok_done(li: int):
            self._announce(red, *self._layer_range("net_token", li))

        dseq = engine.stack_backward(tspec, Wt, self._G["net_token"], ctx_tok, dh,
                                     self.rope("net_token"), accumulate, tok_done)
        del dh, ctx_tok
        acc32 = torch.zeros((V, D), dtype=torch.float32, device=dev)
        src, seg = ops.token_segments(y_t[:, : T - 1], V, row_mul=T, col_mul=1, add=1)  # row of dseq per occurrence
        ops.embed_segment_bwd(src, seg, dseq, D, acc32, tok.pad_id)
        ops.cast_from_f32(acc32, self._G["net_token"].embed, accumulate)
        o, n, _ = self._offsets["net_token.embed_tokens.weight"]
        self._announce(red, o, o + n)
        dhid_t = torch.empty((N, D), dtype=dty, device=dev)
        ops.copy_rows(dseq, T * D, dhid_t, D, N, D)
        del dseq
        if sel is not None:
            dhidden = torch.zeros((B, S, D), dtype=dty, device=dev)
            dhidden.index_add_(1, sel, dhid_t.view(B, -1, D))
            dhidden = dhidden.view(M, D)
        else:
            dhidden = dhid_t

        def net_done(li: int):
            self._announce(red, *self._layer_range("net", li))

        dx = engine.stack_backward(spec, Wn, self._G["net"], ctx_net, dhidden, self.rope("net"),
                                   accumulate, net_done)
        acc32.zero_()
        src, seg = ops.token_segments(x.view(-1, T), V, row_mul=1, col_mul=0, add=0)  # every token of an event reads the event's row
        ops.embed_segment_bwd(src, seg, dx, D, acc32, tok.pad_id)
        ops.cast_from_f32(acc32, self._G["net"].embed, accumulate)
        o, n, _ = self._offsets["net.embed_tokens.weight"]
        self._announce(red, o, o + n)
        self._announce(red, self._n_mat, self._flat.numel())  # all norm vectors
        self._micro += 1
        return loss, acc

    def _layer_range(self, pre: str, li: int) -> Tuple[int, int]:
        lo = self._offsets[f"{pre}.layers.{li}.self_attn.q_proj.weight"][0]
        o, n, _ = self._offsets[f"{pre}.layers.{li}.mlp.down_proj.weight"]
        return lo, o + n

    def use_comm(self, comm) -> "TrainMIDIModel":
        """exchange gradients (and broadcast parameters) through ``comm`` (an ``MHComm``) instead of torch.distributed"""
        self.comm = comm
        self._reducer = None
        return self

    def _world(self) -> int:
        import torch.distributed as dist
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.process_group) if dist.is_available() and dist.is_initialized() else 1

    def _reducer_for_step(self):
        if self._world() == 1 and not self.force_reduce:
            return None
        if self._lora is not None:  # only the adapter gradients are exchanged (in optimizer_step)
            return None
        # exchange only on the micro-batch that completes an accumulation window (DDP no_sync otherwise)
        if (self._micro + 1) % max(1, self.accumulate_grad_batches) != 0:
            return None
        r = self._reducer
        if r is None or r.flat is not self._flat_grad or r.comm is not self.comm or r.force != self.force_reduce:
            self._reducer = GradReducer(self._flat_grad, self.process_group, self.bucket_mb << 20, comm=self.comm,
                                        force=self.force_reduce)
        return self._reducer

    @staticmethod
    def _announce(red, lo: int, hi: int):
        if red is not None:
            red.ready(lo, hi)

    # ------------------------------------------------------------------------- reference-named API
    def training_step(self, batch, batch_idx: int = 0):
        loss, _ = self._loss_and_backward(batch, backward=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0):
        """train.py:190-206: returns (loss, acc); with a process group the two scalars are averaged over ranks
        (``sync_dist=True``)."""
        loss, acc = self._loss_and_backward(batch, backward=False, want_acc=True)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            pair = torch.stack([loss.reshape(()), acc.reshape(())])
            dist.all_reduce(pair, group=self.process_group)
            pair /= dist.get_world_size(self.process_group)
            loss, acc = pair[0:1], pair[1]
        return loss, acc

    def compute_accuracy(self, logits, labels):
        """train.py:153-166 on materialised logits (API compatibility; the fused path gets argmax from the CE kernel)."""
        out = torch.argmax(logits, dim=-1).flatten()
        labels = labels.flatten()
        mask = labels != self.tokenizer.pad_id
        return (out[mask] == labels[mask]).float().sum() / mask.sum()

    def fit_step(self, batch):
        """One micro-batch + (on the accumulation boundary) one optimiser step; returns the loss tensor."""
        loss = self.training_step(batch)
        if self._micro % max(1, self.accumulate_grad_batches) == 0:
            self.optimizer_step()
        return loss

    def broadcast_parameters(self, src: int = 0):
        """DDP constructor behaviour: rank `src`'s weights everywhere (one flat broadcast)."""
        import torch.distributed as dist
        if self.comm is not None:
            if self.comm.world > 1 or self.force_reduce:
                self.comm.broadcast_(self._flat, root=src)
        elif dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            dist.broadcast(self._flat, src=src, group=self.process_group)
        self.weights_written()
