"""``MIDIModel`` with the reference's surface (midi_model.py:99-250) on the HIP kernels.

Same constructor, attributes (``tokenizer, net, net_token, lm_head, config, device, dtype``), the same
140 parameter names/shapes (so ``state_dict`` / ``load_state_dict`` / ``.to()`` interoperate with
reference checkpoints), and the same methods: ``forward``, ``forward_token``, ``sample_top_p_k``,
``generate``.  Underneath, all parameters live in ONE flat buffer (q|k|v and gate|up adjacent, so each pair
is one projection; gradients mirror the layout so the data-parallel reducer and the fused AdamW work on
contiguous ranges), and every tensor op is a launch from ``engine.py`` / ``ops.py``.

There is no CPU implementation: calling the compute methods without a ROCm device raises.
"""
from __future__ import annotations

import json
import math
import os
import threading
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import engine, ops
from .config import MIDIModelConfig, NetConfig
from .engine import KVState, LayerTensors, RopeTable, StackSpec, StackTensors


# MH_DECODE_SPEC=1: queue the net step of an event BEFORE the host has read its tokens (decode.DecodeSession.sample_event,
# tokens through a copy stream).  Measured slower on the MI355X, same box, interleaved (r02, tools/gpu_r2_13.sh): 1.86 ms per
# event against 1.70 with the plain order (token steps -> host copy -> noise graph + net step), pinned or pageable copy
# alike -- the host's ~40 us round trip is cheaper than what the extra queue traffic costs the dependent chain.  Off.
_SPECULATIVE_NET = os.environ.get("MH_DECODE_SPEC", "0") == "1"


class _SessionPool:
    """Idle decode sessions of one model (buffers + captured graphs).  Copies and pickles of the model start empty."""

    def __init__(self):
        self.lock = threading.Lock()
        self.idle = []

    def __deepcopy__(self, memo):
        return _SessionPool()

    def __reduce__(self):
        return (_SessionPool, ())


class _W(nn.Module):
    """A module that owns one ``weight`` (stands in for nn.Linear / nn.Embedding / RMSNorm in the key tree)."""

    def __init__(self, *shape: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))


class _Attn(nn.Module):
    def __init__(self, D: int):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = _W(D, D), _W(D, D), _W(D, D), _W(D, D)


class _MLP(nn.Module):
    def __init__(self, D: int, I: int):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = _W(I, D), _W(I, D), _W(D, I)


class _Layer(nn.Module):
    def __init__(self, D: int, I: int):
        super().__init__()
        self.self_attn = _Attn(D)
        self.mlp = _MLP(D, I)
        self.input_layernorm = _W(D)
        self.post_attention_layernorm = _W(D)


class _Stack(nn.Module):
    def __init__(self, cfg: NetConfig):
        super().__init__()
        self.config = cfg
        D, I = cfg.hidden_size, cfg.intermediate_size
        self.embed_tokens = _W(cfg.vocab_size, D)
        self.layers = nn.ModuleList([_Layer(D, I) for _ in range(cfg.num_hidden_layers)])
        self.norm = _W(D)


_MATS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
         "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


class MIDIModel(nn.Module):
    config_class = MIDIModelConfig

    def __init__(self, config: MIDIModelConfig, *args, **kwargs):
        super().__init__()
        self.config = config
        self.tokenizer = config.tokenizer
        self.net = _Stack(config.net_config)
        self.net_token = _Stack(config.net_token_config)
        self.lm_head = _W(self.tokenizer.vocab_size, config.n_embd)
        self._specs = {
            "net": self._spec("net", config.net_config, "event"),
            "net_token": self._spec("net_token", config.net_token_config, "token"),
        }
        self._flat = None
        self._flat_grad = None
        self._ropes = {}
        self._tables = None
        self._sessions = _SessionPool()  # idle decode sessions (decode.py)
        self._weights_epoch = 0          # bumped by every writer that bypasses torch's version counters (weights_written)
        self._fold_cache = {}            # stack name -> (key, folded weights): folded_weights()
        self._reset_parameters()
        self._repack()

    # ------------------------------------------------------------------------------------------ set-up
    @staticmethod
    def _spec(name: str, c: NetConfig, kind: str) -> StackSpec:
        return StackSpec(name, c.hidden_size, c.num_attention_heads, c.intermediate_size, c.num_hidden_layers,
                         float(c.rms_norm_eps), float(c.rope_theta), kind)

    def _reset_parameters(self):
        """HF LLaMA init for the two stacks (N(0, 0.02), pad row zero, norms one; modeling_llama post_init) and
        torch's default nn.Linear init for lm_head (the reference never calls post_init on MIDIModel itself)."""
        with torch.no_grad():
            for st in (self.net, self.net_token):
                for n, p in st.named_parameters():
                    if p.dim() == 1:
                        p.fill_(1.0)
                    else:
                        p.normal_(0.0, 0.02)
                pad = st.config.pad_token_id
                if pad is not None:
                    st.embed_tokens.weight[pad].zero_()
            bound = 1.0 / math.sqrt(self.lm_head.weight.shape[1])
            self.lm_head.weight.uniform_(-bound, bound)

    def _layout(self) -> List[Tuple[str, nn.Parameter, str]]:
        """Flat order: all matrices in forward order, then all norm vectors (the no-weight-decay group)."""
        named = dict(self.named_parameters())
        mats, norms = [], []
        for pre, st in (("net", self.net), ("net_token", self.net_token)):
            mats.append(f"{pre}.embed_tokens.weight")
            for i in range(len(st.layers)):
                mats += [f"{pre}.layers.{i}.{m}.weight" for m in _MATS]
                norms += [f"{pre}.layers.{i}.input_layernorm.weight", f"{pre}.layers.{i}.post_attention_layernorm.weight"]
            norms.append(f"{pre}.norm.weight")
        mats.append("lm_head.weight")
        assert len(mats) + len(norms) == len(named)
        return [(n, named[n], "mat") for n in mats] + [(n, named[n], "norm") for n in norms]

    def _repack(self):
        """(Re)build the flat parameter buffer on the parameters' current device/dtype and re-point every
        Parameter at its slice.  Runs after construction and after every ``.to()/.cuda()/.bfloat16()``."""
        layout = self._layout()
        p0 = layout[0][1]
        total = sum(p.numel() for _, p, _ in layout)
        flat = torch.empty(total, dtype=p0.dtype, device=p0.device)
        off = 0
        self._offsets = {}
        with torch.no_grad():
            for name, p, grp in layout:
                n = p.numel()
                assert off % 8 == 0, "every tensor must start 16-byte aligned"
                flat[off:off + n].copy_(p.detach().reshape(-1).to(flat.dtype))
                p.data = flat[off:off + n].view(p.shape)
                p.grad = None
                self._offsets[name] = (off, n, grp)
                off += n
        self._flat = flat
        self._n_mat = sum(n for (_, n, g) in self._offsets.values() if g == "mat")
        self._flat_grad = None
        self._ropes = {}
        self._tables = None
        self._sessions = _SessionPool()  # idle decode sessions (decode.py)
        self._W = {k: self._stack_views(k, flat) for k in ("net", "net_token")}
        self._fold_cache = {}
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1

    # --------------------------------------------------------------------- derived data of the weights
    def weights_written(self) -> None:
        """Tell the model that its flat parameter buffer was written through a raw pointer (mh_adamw, a LoRA materialisation, a
        broadcast): torch's version counters do not see those writes, the caches of data DERIVED from the weights
        (folded_weights) are keyed on this epoch as well."""
        self._weights_epoch += 1

    def folded_weights(self, pre: str = "net"):
        """engine.fold_norm_weights of stack ``pre`` -- [(wqkv * n1, wgu * n2) per layer], the weights the forward-only blocks with
        folded RMSNorms multiply by (engine.layer_forward_folded) -- kept on the model and re-derived lazily, in place, when a
        parameter changed: keyed on the parameters' version counters (torch-side in-place writes: load_state_dict, an optimizer
        from torch.optim, ``p.mul_()`` under no_grad) and on the epoch raw-pointer writers bump (weights_written; a write through
        ``p.data`` is such a writer too -- torch does not count it).  So the public
        ``forward`` under no_grad reaches the folded blocks from 8192 rows up without paying the fold per call (~0.7 ms of
        elementwise launches for tv2o-medium).  277 MB for tv2o-medium in bf16."""
        st = getattr(self, pre)
        versions = tuple(p._version for lyr in st.layers for p in lyr.parameters())
        key = (self._weights_epoch, self._flat.data_ptr(), versions)
        hit = self._fold_cache.get(pre)
        if hit is not None and hit[0] == key:
            return hit[1]
        with self._sessions.lock:  # (app.py:496 runs up to ten generators on one model: one of them derives, the others wait)
            hit = self._fold_cache.get(pre)
            if hit is None or hit[0] != key:
                with torch.no_grad():
                    folded = engine.fold_norm_weights(self._W[pre], out=hit[1] if hit is not None else None)
                self._fold_cache[pre] = hit = (key, folded)
        return hit[1]

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if getattr(self, "_flat", None) is not None or hasattr(self, "_offsets"):
            self._repack()
        return out

    def _stack_views(self, pre: str, flat: torch.Tensor) -> StackTensors:
        spec = self._specs[pre]
        D, I = spec.D, spec.I

        def view(name, rows, cols=None):
            off, _, _ = self._offsets[name]
            n = rows * (cols or 1)
            t = flat[off:off + n]
            return t.view(rows, cols) if cols else t

        V = self.tokenizer.vocab_size
        st = StackTensors(embed=view(f"{pre}.embed_tokens.weight", V, D), norm=view(f"{pre}.norm.weight", D))
        for i in range(spec.L):
            b = f"{pre}.layers.{i}."
            st.layers.append(LayerTensors(
                wqkv=view(b + "self_attn.q_proj.weight", 3 * D, D), wo=view(b + "self_attn.o_proj.weight", D, D),
                wgu=view(b + "mlp.gate_proj.weight", 2 * I, D), wd=view(b + "mlp.down_proj.weight", D, I),
                n1=view(b + "input_layernorm.weight", D), n2=view(b + "post_attention_layernorm.weight", D)))
        return st

    # ------------------------------------------------------------------------------------- properties
    @property
    def device(self) -> torch.device:
        return self._flat.device

    @property
    def dtype(self) -> torch.dtype:
        return self._flat.dtype

    @property
    def vocab_padded(self) -> int:
        return ops.round_up(self.tokenizer.vocab_size, 64)

    def _require_gpu(self):
        if self._flat.device.type != "cuda":
            raise RuntimeError("MIDIModel compute needs a ROCm device: move the model with .to('cuda'). "
                               "This implementation has no CPU path (the reference's CPU path is the oracle's job).")
        if self._flat.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError(f"unsupported parameter dtype {self._flat.dtype} (float32 or bfloat16)")

    def _check_ids(self, ids: torch.Tensor) -> None:
        """Token ids still on the host are range-checked for free; ids already on the device are checked by the embedding
        kernels themselves (an id outside [0, vocab) traps the kernel, as torch's embedding raises a device assert)."""
        if not ids.is_cuda and ids.numel():
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= self.tokenizer.vocab_size:
                raise IndexError(f"token ids outside [0, {self.tokenizer.vocab_size}): min {lo}, max {hi}")

    def rope(self, pre: str) -> RopeTable:
        if pre not in self._ropes:
            spec = self._specs[pre]
            self._ropes[pre] = RopeTable(spec.hd, spec.theta, self.device)
        return self._ropes[pre]

    # ---------------------------------------------------------------- gradients + transposed weights
    def grad_buffer(self) -> torch.Tensor:
        """Flat gradient buffer mirroring the parameter layout; every p.grad is a view of it."""
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self._flat)
            for name, p in self.named_parameters():
                off, n, _ = self._offsets[name]
                p.grad = self._flat_grad[off:off + n].view(p.shape)
            self._G = {k: self._stack_views(k, self._flat_grad) for k in ("net", "net_token")}
            off, n, _ = self._offsets["lm_head.weight"]
            self._g_lm = self._flat_grad[off:off + n].view(self.lm_head.weight.shape)
        return self._flat_grad

    # ------------------------------------------------------------------------------ reference methods
    def load_merge_lora(self, model_id):
        """midi_model.py:109-114 without peft: read a saved LoRA adapter (``adapter_config.json`` +
        ``adapter_model.safetensors`` / ``.bin`` in the directory ``model_id``) and merge it into the weights in place,
        W += scale * B @ A with scale = lora_alpha / r (or / sqrt(r) for rsLoRA), which is what
        ``LoraModel.merge_and_unload`` computes for ``nn.Linear`` targets (train.py:439-449 trains r=64, alpha=128 on
        q,k,v,o,gate,up,down).  Returns self.  The merge is a pure weight transform; nothing else changes."""
        cfg_path = os.path.join(model_id, "adapter_config.json")
        if not os.path.isdir(model_id) or not os.path.exists(cfg_path):
            raise FileNotFoundError(f"{model_id}: not a local LoRA adapter directory (hub ids need network access)")
        with open(cfg_path) as f:
            cfg = json.load(f)
        if cfg.get("peft_type", "LORA") != "LORA":
            raise ValueError(f"unsupported peft_type {cfg.get('peft_type')}")
        r, alpha = int(cfg["r"]), float(cfg.get("lora_alpha", cfg["r"]))
        scale = alpha / math.sqrt(r) if cfg.get("use_rslora") else alpha / r
        st_path = os.path.join(model_id, "adapter_model.safetensors")
        if os.path.exists(st_path):
            from safetensors.torch import load_file
            sd = load_file(st_path)
        else:
            sd = torch.load(os.path.join(model_id, "adapter_model.bin"), map_location="cpu")
        params = dict(self.named_parameters())
        merged = 0
        with torch.no_grad():
            for key, a in sd.items():
                if not key.endswith("lora_A.weight"):
                    continue
                stem = key[: -len(".lora_A.weight")]
                b = sd[stem + ".lora_B.weight"]
                name = stem
                for prefix in ("base_model.model.", "base_model."):
                    if name.startswith(prefix):
                        name = name[len(prefix):]
                        break
                w = params.get(name + ".weight")
                if w is None:
                    raise KeyError(f"adapter targets {name}.weight, which this model does not have")
                delta = (b.float() @ a.float()) * scale  # [out, r] @ [r, in]
                if cfg.get("fan_in_fan_out"):
                    delta = delta.t()
                if delta.shape != w.shape:
                    raise ValueError(f"{name}: adapter delta {tuple(delta.shape)} vs weight {tuple(w.shape)}")
                w.add_(delta.to(device=w.device, dtype=w.dtype))
                merged += 1
        if merged == 0:
            raise ValueError(f"{model_id}: no lora_A/lora_B pairs found")
        return self

    def forward(self, x: torch.Tensor, cache=None) -> torch.Tensor:
        """x (B, S, 8) int64 -> hidden (B, S, n_embd)   [midi_model.py:137-150]

        Without ``cache`` the call is differentiable (an autograd node that runs the explicit backward
        schedule).  With ``cache`` (any object; HF DynamicCache instances are accepted) K/V go to
        preallocated buffers attached to it: an empty cache is prefilled causally, afterwards one event per
        call is decoded, or a chunk of S > 1 events is appended in one pass (``engine.stack_extend``)."""
        self._require_gpu()
        if x.dim() != 3:
            raise ValueError(f"expected (batch, events, tokens) ids, got shape {tuple(x.shape)}")
        B, S, T = x.shape
        self._check_ids(x)
        x = x.to(device=self.device, dtype=torch.long).contiguous()
        if cache is None:
            if not torch.is_grad_enabled():
                # nothing will backpropagate (validation, app.py's no_grad calls): the forward-only schedule -- no activations kept,
                # gate|up never written, and for enough rows the folded-norm blocks.  (Inside an autograd.Function the grad mode is
                # always off and needs_input_grad still reports the parameters' flags, so the decision is taken here.)
                spec = self._specs["net"]
                e = torch.empty((B * S, spec.D), dtype=self.dtype, device=self.device)
                ops.embed_sum_fwd(x.view(B * S, T), self._W["net"].embed, e)
                y, _ = engine.stack_forward(spec, self._W["net"], e, B, S, self.rope("net"), save=False,
                                            folded=self._folded_for(spec, e))
                return y.view(B, S, spec.D)
            from .autograd import NetFn
            params = self._stack_params("net")
            return NetFn.apply(self, x, *params)
        spec = self._specs["net"]
        st = getattr(cache, "_mh_state", None)
        with torch.no_grad():
            e = torch.empty((B * S, spec.D), dtype=self.dtype, device=self.device)
            ops.embed_sum_fwd(x.view(B * S, T), self._W["net"].embed, e)
            if st is None or st.len == 0:
                if st is None:
                    st = KVState(spec, B, max(ops.round_up(S + 64, 64), 256), e)
                    cache._mh_state = st
                y = engine.stack_prefill(spec, self._W["net"], e, B, S, self.rope("net"), st, folded=self._folded_for(spec, e))
                return y.view(B, S, spec.D)
            if st.B != B:
                raise ValueError(f"cache was built for batch {st.B}, got {B}")
            if S == 1:  # one event per call: the decode step
                return engine.stack_decode(spec, self._W["net"], e, self.rope("net"), st).view(B, 1, spec.D)
            # chunked continuation: the S new events attend to the cached ones and causally to each other
            return engine.stack_extend(spec, self._W["net"], e, B, S, self.rope("net"), st).view(B, S, spec.D)

    def _folded_for(self, spec, e: torch.Tensor):
        """the kept folded weights when a forward-only pass over ``e`` [rows, D] will run the folded-norm blocks, else None"""
        if e.shape[0] >= engine.FOLD_MIN_ROWS_PREFOLDED and ops.norm_fold_ok(e, spec.D, spec.hd, spec.I):
            return self.folded_weights(spec.name)
        return None

    def forward_token(self, hidden_state=None, x=None, cache=None) -> torch.Tensor:
        """hidden_state (N, n_embd) and/or x (N, t) int64 -> logits (N, [1]+t, vocab)   [midi_model.py:116-135]"""
        self._require_gpu()
        if hidden_state is None and x is None:
            raise ValueError("forward_token needs hidden_state and/or x")
        if x is not None:
            self._check_ids(x)
            x = x.to(device=self.device, dtype=torch.long)
        if cache is None:
            from .autograd import TokFn
            params = self._stack_params("net_token") + [self.lm_head.weight]
            return TokFn.apply(self, hidden_state, x, *params)
        spec = self._specs["net_token"]
        st = getattr(cache, "_mh_state", None)
        V, Vp = self.tokenizer.vocab_size, self.vocab_padded
        with torch.no_grad():
            rows = []
            N = (hidden_state if hidden_state is not None else x).shape[0]
            if hidden_state is not None:
                rows.append(hidden_state.to(self.dtype).contiguous())
            if x is not None:
                for j in range(x.shape[1]):
                    rows.append(self._W["net_token"].embed[x[:, j]])  # gather of N rows (indexing only)
            if st is None:
                st = KVState(spec, N, self.tokenizer.max_token_seq, rows[0])
                cache._mh_state = st
            outs = []
            for r in rows:
                h = engine.stack_decode(spec, self._W["net_token"], r.contiguous(), self.rope("net_token"), st)
                lg = torch.empty((N, Vp), dtype=self.dtype, device=self.device)
                ops.gemm_nt(h, self.lm_head.weight.data, lg[:, :V])
                outs.append(lg[:, :V])
            return torch.stack(outs, dim=1)

    def _stack_params(self, pre: str) -> List[nn.Parameter]:
        st = getattr(self, pre)
        out = [st.embed_tokens.weight]
        for l in st.layers:
            out += [l.self_attn.q_proj.weight, l.self_attn.k_proj.weight, l.self_attn.v_proj.weight,
                    l.self_attn.o_proj.weight, l.mlp.gate_proj.weight, l.mlp.up_proj.weight, l.mlp.down_proj.weight,
                    l.input_layernorm.weight, l.post_attention_layernorm.weight]
        out.append(st.norm.weight)
        return out

    def sample_top_p_k(self, probs, p, k, generator=None):
        """midi_model.py:152-165, op for op (torch.sort / cumsum / multinomial), so a seeded
        ``torch.Generator`` is consumed exactly as the reference consumes it."""
        probs_sort, probs_idx = torch.sort(probs, dim=-1, descending=True)
        probs_sum = torch.cumsum(probs_sort, dim=-1)
        mask = probs_sum - probs_sort > p
        probs_sort[mask] = 0.0
        mask = torch.zeros(probs_sort.shape[-1], device=probs_sort.device)
        mask[:k] = 1
        probs_sort = probs_sort * mask
        probs_sort.div_(probs_sort.sum(dim=-1, keepdim=True))
        shape = probs_sort.shape
        next_token = torch.multinomial(probs_sort.reshape(-1, shape[-1]), num_samples=1,
                                       generator=generator).reshape(*shape[:-1], 1)
        return torch.gather(probs_idx, -1, next_token).reshape(*shape[:-1])

    # ------------------------------------------------------------------------------------- generation
    def _grammar(self):
        if self._tables is None:
            from .tokenizer import grammar_tables
            first, lo, hi, arity = grammar_tables(self.tokenizer)
            dev = self.device
            self._tables = (torch.tensor(first, dtype=torch.uint8, device=dev),
                            torch.tensor(lo, dtype=torch.int32, device=dev),
                            torch.tensor(hi, dtype=torch.int32, device=dev), arity)
        return self._tables

    def generate(self, prompt=None, batch_size=1, max_len=512, temp=1.0, top_p=0.98, top_k=20, generator=None,
                 ban_eos: bool = False, disable_patch_change: bool = False, disable_control_change: bool = False,
                 disable_channels=None):
        """midi_model.py:167-250 with the host-bound parts moved to the device: grammar masks come from
        per-event-id range tables (no Python loop over the batch), K/V live in preallocated buffers, every decode /
        sampling step is a replayed hipGraph (decode.py) and the only device->host traffic is ONE copy of the B sampled
        event ids per event (the reference does B ``.item()`` calls).  The number of sampling calls per event follows
        the reference's break rule exactly, so a seeded generator yields the reference's stream.

        Returns ``np.ndarray (B, <= max_len, 8) int64`` including the prompt.  Extras (ours): ``ban_eos`` removes EOS
        from the first-token mask (throughput runs); ``disable_patch_change`` / ``disable_control_change`` /
        ``disable_channels`` are the mask options of the serving loop (app.py:27-31, 73-86)."""
        inp = self._prompt_tensor(prompt, batch_size)
        parts = [inp.cpu().numpy()]
        for ev in self._generate_events(inp, batch_size, max_len, temp, top_p, top_k, generator, ban_eos,
                                        disable_patch_change, disable_control_change, disable_channels):
            parts.append(ev[:, None, :])
        return np.concatenate(parts, axis=1)

    def generate_stream(self, prompt=None, batch_size=1, max_len=512, temp=1.0, top_p=0.98, top_k=20,
                        disable_patch_change=False, disable_control_change=False, disable_channels=None, generator=None):
        """The serving loop of the reference (app.py:27-120, same arguments): a Python generator that yields every new
        event as ``np.ndarray (B, 8) int64`` as soon as it is sampled; the prompt is cropped to its last 4096 events
        (app.py:53).  Same kernels and sessions as ``generate``."""
        inp = self._prompt_tensor(prompt, batch_size)[:, -4096:]
        return self._generate_events(inp, batch_size, max_len, temp, top_p, top_k, generator, False,
                                     disable_patch_change, disable_control_change, disable_channels)

    def _prompt_tensor(self, prompt, batch_size: int) -> torch.Tensor:
        """prompt handling of midi_model.py:171-188 (and app.py:35-52): None -> one BOS event per sequence"""
        self._require_gpu()
        tok = self.tokenizer
        T, dev = tok.max_token_seq, self.device
        if prompt is None:
            inp = torch.full((batch_size, 1, T), tok.pad_id, dtype=torch.long, device=dev)
            inp[:, 0, 0] = tok.bos_id
            return inp
        prompt = np.asarray(prompt)
        if prompt.size and (prompt.min() < 0 or prompt.max() >= tok.vocab_size):
            raise ValueError(f"prompt holds token ids outside [0, {tok.vocab_size}): min {prompt.min()}, max {prompt.max()}")
        if prompt.ndim == 2:
            prompt = np.repeat(prompt[None, :], repeats=batch_size, axis=0)
        elif prompt.shape[0] == 1:
            prompt = np.repeat(prompt, repeats=batch_size, axis=0)
        elif prompt.ndim != 3 or prompt.shape[0] != batch_size:
            raise ValueError(f"invalid shape for prompt, {prompt.shape}")
        prompt = prompt[..., :T]
        if prompt.shape[-1] < T:
            prompt = np.pad(prompt, ((0, 0), (0, 0), (0, T - prompt.shape[-1])), mode="constant",
                            constant_values=tok.pad_id)
        return torch.from_numpy(np.ascontiguousarray(prompt)).to(dtype=torch.long, device=dev)

    def _generate_events(self, inp, batch_size, max_len, temp, top_p, top_k, generator, ban_eos, disable_patch_change,
                         disable_control_change, disable_channels):
        """generator over the new events ((B, 8) int64 numpy each); the session is returned to the pool when the
        generator finishes or is closed"""
        tok = self.tokenizer
        T = tok.max_token_seq
        B = batch_size
        cur_len = inp.shape[1]
        if cur_len >= max_len:
            return
        # Every grammar mask must keep at least one id.  The reference fails with a (recoverable) multinomial error when a
        # mask is empty (midi_model.py:152-165 on an all-zero row); the fused sampler has no such path -- it would hand a
        # sentinel id to the next embedding lookup, which traps -- so an emptied mask is refused here, before any launch.
        first_ids = {tok.eos_id, *tok.event_ids.values()}
        if ban_eos:
            first_ids.discard(tok.eos_id)
        if disable_patch_change:
            first_ids.discard(tok.event_ids["patch_change"])
        if disable_control_change:
            first_ids.discard(tok.event_ids["control_change"])
        if not first_ids:
            raise ValueError("generate: the options leave no legal event id at token position 0")
        chans = sorted(set(int(c) for c in (disable_channels or [])))
        n_chan = len(tok.parameter_ids["channel"])
        if any(c < 0 or c >= n_chan for c in chans):
            raise ValueError(f"generate: disable_channels holds a channel outside [0, {n_chan}): {chans}")
        if len(chans) >= n_chan:
            raise ValueError("generate: disable_channels bans every channel id: no legal token is left at the channel position")
        with torch.inference_mode():
            ses = self._checkout_session(B, max(max_len, cur_len) + 1, float(temp), float(top_p), int(top_k))
        try:
            with torch.inference_mode():
                ses.first_mask.copy_(self._grammar()[0])
                if ban_eos:
                    ses.first_mask[tok.eos_id] = 0
                if disable_patch_change:
                    ses.first_mask[tok.event_ids["patch_change"]] = 0
                if disable_control_change:
                    ses.first_mask[tok.event_ids["control_change"]] = 0
                ses.ban.zero_()
                for c in chans:
                    ses.ban[tok.parameter_ids["channel"][c]] = 1
                ses.reset()
                ses.begin(generator)
                ses.prefill(inp)  # causal forward over the prompt; hidden = last position
            while cur_len < max_len:
                with torch.inference_mode():
                    more = cur_len + 1 < max_len
                    speculative = more and ses.g_steps is not None and _SPECULATIVE_NET
                    # the event's 8 token steps (one replayed graph, one host copy); in the graph form the net step over
                    # the event is queued behind them before the host waits
                    event, end_all = ses.sample_event(then_net=speculative)
                    cur_len += 1
                    last = end_all or not more
                    if not last:
                        ses.draw_noise()  # the next event's draws, on a side stream under the net step
                        if not speculative:
                            ses.net_step()  # decode the event just sampled; hidden = its net output (overlaps the consumer)
                yield event
                if last:
                    break
        finally:
            ses.end()
            self._return_session(ses)

    # decode sessions: buffers + captured graphs, one per concurrent generate() call (decode.py)
    def _checkout_session(self, B: int, need: int, temp: float, top_p: float, top_k: int):
        from .decode import DecodeSession
        cap = 256
        while cap < need:
            cap *= 2
        key = DecodeSession.make_key(self, B, cap, temp, top_p, top_k)
        pool = self._sessions
        with pool.lock:
            found = None
            for k, ses in enumerate(pool.idle):
                if ses.key == key:
                    found = pool.idle.pop(k)
                    break
            while found is None and len(pool.idle) >= 4:  # bound the memory held by idle sessions
                pool.idle.pop(0)
        if found is not None:
            found.refresh()  # weights may have been trained / merged since the session derived its folded copies
            return found
        return DecodeSession(self, B, cap, temp, top_p, top_k)

    def _return_session(self, ses) -> None:
        with self._sessions.lock:
            self._sessions.idle.append(ses)

    # ------------------------------------------------------------------------------------ persistence
    def save_pretrained(self, save_directory: str):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "model.safetensors"))

    @classmethod
    def from_pretrained(cls, directory: str):
        from safetensors.torch import load_file
        cfg = MIDIModelConfig.from_json_file(os.path.join(directory, "config.json"))
        model = cls(cfg)
        model.load_checkpoint_state(load_file(os.path.join(directory, "model.safetensors")))
        return model

    # keys a reference-side checkpoint may carry that are not parameters of the model (buffers of older HF versions)
    _BENIGN_EXTRA = ("rotary_emb.inv_freq",)

    def load_checkpoint_state(self, state: dict):
        """Load a reference checkpoint's tensors the way app.py:311-316 / train.py:433-436 do (``strict=False``) but
        WITHOUT its silence: a Lightning ``.ckpt`` payload (``{"state_dict": ...}``) is unwrapped, the prefixes other
        wrappers add (``model.``, ``base_model.model.``, ``module.``) are stripped when they cover every key, known-benign
        extras are dropped, and any parameter the file does not supply -- or any other unexpected key -- raises instead of
        leaving random-initialised weights behind.  Returns self."""
        if "state_dict" in state and not torch.is_tensor(state["state_dict"]):
            state = state["state_dict"]
        own = set(self.state_dict().keys())
        # wrapper prefixes in any order and nesting (torch.compile's _orig_mod., DDP's module., Lightning's model., peft's
        # base_model.model.): stripped, one at a time, until none covers every key or the keys are ours
        prefixes = ("model.", "base_model.model.", "module.", "_orig_mod.")
        for _ in range(8):
            if own & set(state):
                break
            hit = next((p for p in prefixes if state and all(k.startswith(p) for k in state)), None)
            if hit is None:
                break
            state = {k[len(hit):]: v for k, v in state.items()}
        if any(".base_layer." in k or ".lora_A." in k or ".lora_B." in k for k in state):
            raise RuntimeError("this checkpoint holds peft-wrapped modules (base_layer / lora_A / lora_B keys): load the base weights "
                               "with load_checkpoint_state and the adapter with MIDIModel.load_merge_lora(adapter_dir), or merge "
                               "the adapter before saving (merge_and_unload)")
        state = {k: v for k, v in state.items() if not k.endswith(self._BENIGN_EXTRA)}
        res = self.load_state_dict(state, strict=False)
        if res.missing_keys or res.unexpected_keys:
            raise RuntimeError(f"checkpoint does not match the model: missing {sorted(res.missing_keys)[:8]} "
                               f"({len(res.missing_keys)}), unexpected {sorted(res.unexpected_keys)[:8]} "
                               f"({len(res.unexpected_keys)})")
        return self

    @classmethod
    def from_checkpoint(cls, config, path: str, **kwargs):
        """app.py:304-316: a Lightning ``.ckpt`` (torch pickle with ``state_dict``) or a ``.safetensors`` file next to a
        config given by name / object."""
        cfg = config if isinstance(config, MIDIModelConfig) else MIDIModelConfig.from_name(config)
        trust = bool(kwargs.pop("trust_checkpoint", False))
        model = cls(cfg, **kwargs)
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            state = load_file(path)
        else:
            import pickle
            try:
                state = torch.load(path, map_location="cpu", weights_only=True)
            except pickle.UnpicklingError as e:
                # the reference calls plain torch.load (app.py:314), which executes whatever the pickle names; a Lightning
                # .ckpt with callback / hyper-parameter objects outside torch's allow-list lands here
                # the trust is per call: the caller vouches for THIS file (no process-wide switch)
                if not trust:
                    raise RuntimeError(f"{path} holds pickled objects beyond tensors ({e}); re-save its state_dict alone, or pass "
                                       "from_checkpoint(..., trust_checkpoint=True) to unpickle it as the reference does (only "
                                       "for a file you trust: unpickling executes what the file names)") from e
                import warnings
                warnings.warn(f"from_checkpoint: unpickling {path} with weights_only=False (trust_checkpoint=True)", stacklevel=2)
                state = torch.load(path, map_location="cpu", weights_only=False)
        return model.load_checkpoint_state(state)
