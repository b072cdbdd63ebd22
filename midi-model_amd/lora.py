"""LoRA fine-tuning on the fused training step (train.py:439-449: ``--task lora``: frozen base model, rank-64 adapters
with alpha 128 and dropout 0 on q, k, v, o, gate, up, down of every decoder layer, ``model.add_adapter(lora_config)``).

With dropout 0 the adapted layer is exactly a Linear with weight  W_eff = W + (alpha / r) * B @ A,  so the step keeps
running on the same kernels:

  * before a step the effective weights are materialised into the live parameter buffer
    (one rank-r GEMM per target: W_eff = base + s * B @ A, 105 targets of tv2o-medium ~ 1 ms);
  * the backward produces d W_eff for every target exactly as in full fine-tuning (same wgrad launches);
  * the adapter gradients follow by the chain rule,  dB = s * dW_eff @ A^T,  dA = s * B^T @ dW_eff  (two rank-r GEMMs
    per target), computed once per optimiser step from the accumulated d W_eff;
  * clip + AdamW run over the flat adapter buffer only; data parallelism all-reduces that buffer (27 MB for r = 64)
    instead of the full gradient.

Nothing of ``peft`` is needed; adapters are saved in its on-disk format (``adapter_config.json`` +
``adapter_model.safetensors``, keys ``base_model.model.<module>.lora_A|lora_B.weight``), which
``MIDIModel.load_merge_lora`` (midi_model.py:109-114) and peft itself read back.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

DEFAULT_TARGETS = ("q_proj", "o_proj", "k_proj", "v_proj", "gate_proj", "up_proj", "down_proj")  # train.py:443


class LoraAdapter:
    def __init__(self, model, r: int = 64, lora_alpha: float = 128.0, target_modules: Sequence[str] = DEFAULT_TARGETS,
                 lora_dropout: float = 0.0, generator: Optional[torch.Generator] = None):
        if lora_dropout != 0.0:
            raise NotImplementedError("LoRA dropout is not supported (the reference trains with lora_dropout=0, "
                                      "train.py:447); with dropout the adapted layer is not a single Linear")
        if r <= 0 or r % 8 != 0:
            raise ValueError(f"LoRA rank must be a positive multiple of 8 (got {r}): adapter rows are 16-byte aligned")
        self.r, self.alpha, self.scale = int(r), float(lora_alpha), float(lora_alpha) / r
        self.target_modules = tuple(target_modules)
        flat = model._flat
        self.targets: List[Tuple[str, int, int, int]] = []  # (module name, out, in, offset of its weight in the flat buffer)
        for name, p in model.named_parameters():
            if name.endswith(".weight") and p.dim() == 2 and name[:-7].split(".")[-1] in self.target_modules \
                    and ".layers." in name:
                off, n, _ = model._offsets[name]
                self.targets.append((name[:-7], p.shape[0], p.shape[1], off))
        if not self.targets:
            raise ValueError(f"no Linear layer matches target_modules={self.target_modules}")
        total = sum(self.r * i + o * self.r for _, o, i, _ in self.targets)
        self.flat = torch.zeros(total, dtype=flat.dtype, device=flat.device)
        self.grad = torch.zeros_like(self.flat)
        self.A: Dict[str, torch.Tensor] = {}
        self.B: Dict[str, torch.Tensor] = {}
        self.gA: Dict[str, torch.Tensor] = {}
        self.gB: Dict[str, torch.Tensor] = {}
        off = 0
        for name, o, i, _ in self.targets:
            for store, gstore, shape in ((self.A, self.gA, (self.r, i)), (self.B, self.gB, (o, self.r))):
                n = shape[0] * shape[1]
                store[name] = self.flat[off:off + n].view(shape)
                gstore[name] = self.grad[off:off + n].view(shape)
                off += n
        # peft's LoRA init: A ~ kaiming_uniform(a = sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)), B = 0 (the adapter starts as a no-op)
        for name, o, i, _ in self.targets:
            bound = 1.0 / math.sqrt(i)
            a = torch.empty((self.r, i), dtype=torch.float32).uniform_(-bound, bound, generator=generator)
            self.A[name].copy_(a.to(device=flat.device, dtype=flat.dtype))
        self.base = flat[: model._n_mat].clone()  # frozen weights (matrix region); the live buffer holds W_eff
        self.opt = None
        self.dirty = True  # live weights do not reflect (A, B) yet

    # -------------------------------------------------------------------------------------------------
    def materialize(self, model) -> None:
        """live weight of every target <- base + scale * B @ A"""
        flat = model._flat
        for name, o, i, off in self.targets:
            w = flat[off:off + o * i].view(o, i)
            ops.gemm_nt(self.B[name], self.A[name], w, K=self.r, alpha=self.scale, beta=1.0,
                        res=self.base[off:off + o * i].view(o, i), tb=True)
        self.dirty = False
        if hasattr(model, "weights_written"):
            model.weights_written()  # (kernel writes into the flat buffer: derived data of the weights is stale)

    def compute_grads(self, model) -> None:
        """adapter gradients from the accumulated gradient of the effective weights"""
        g = model.grad_buffer()
        for name, o, i, off in self.targets:
            dw = g[off:off + o * i].view(o, i)
            ops.gemm_nt(dw, self.A[name], self.gB[name], K=i, alpha=self.scale)                      # dB = s dW A^T
            ops.gemm_nt(self.B[name], dw, self.gA[name], K=o, alpha=self.scale, ta=True, tb=True)    # dA = s B^T dW

    def optimizer_state(self):
        if self.opt is None:
            dev = self.flat.device
            self.opt = {"m": torch.zeros_like(self.flat), "v": torch.zeros_like(self.flat),
                        "sumsq": torch.zeros(1, dtype=torch.float32, device=dev),
                        "partial": torch.empty(1024, dtype=torch.float32, device=dev),
                        "coef": torch.ones(1, dtype=torch.float32, device=dev),
                        "norm": torch.zeros(1, dtype=torch.float32, device=dev)}
        return self.opt

    # -------------------------------------------------------------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {}
        for name, _, _, _ in self.targets:
            sd[f"base_model.model.{name}.lora_A.weight"] = self.A[name].detach().cpu().contiguous()
            sd[f"base_model.model.{name}.lora_B.weight"] = self.B[name].detach().cpu().contiguous()
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        for name, _, _, _ in self.targets:
            for store, tag in ((self.A, "lora_A"), (self.B, "lora_B")):
                key = f"base_model.model.{name}.{tag}.weight"
                alt = f"{name}.{tag}.weight"
                t = sd[key] if key in sd else sd[alt]
                store[name].copy_(t.to(device=self.flat.device, dtype=self.flat.dtype))
        self.dirty = True

    def save(self, directory: str) -> None:
        from safetensors.torch import save_file
        os.makedirs(directory, exist_ok=True)
        save_file(self.state_dict(), os.path.join(directory, "adapter_model.safetensors"))
        cfg = {"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": self.r, "lora_alpha": self.alpha, "lora_dropout": 0.0,
               "bias": "none", "target_modules": list(self.target_modules), "fan_in_fan_out": False, "use_rslora": False}
        with open(os.path.join(directory, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
