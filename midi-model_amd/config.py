"""Model configuration with the reference's surface (midi_model.py:14-96).

``MIDIModelConfig`` keeps ``from_name / get_config / from_json_file / save_pretrained / to_dict``
and the attributes ``tokenizer, net_config, net_token_config, n_embd``.  The two sub-configs are
``NetConfig`` objects that carry the LlamaConfig field names the reference JSON uses
(hidden_size, num_attention_heads, ...), so a ``config.json`` written by either side loads in
the other.  HF ``LlamaConfig`` objects are accepted too (anything with ``to_dict``).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Union

from .tokenizer import MIDITokenizer

config_name_list = ["tv1-medium", "tv2-medium", "tv2o-medium", "tv2-large", "tv2o-large"]


class NetConfig:
    """The LlamaConfig subset that changes the arithmetic (HF defaults)."""

    _fields = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                   num_attention_heads=32, num_key_value_heads=None, hidden_act="silu",
                   max_position_embeddings=2048, rms_norm_eps=1e-6, rope_theta=10000.0,
                   pad_token_id=None, use_cache=True, attention_bias=False, mlp_bias=False,
                   head_dim=None)

    def __init__(self, **kw: Any) -> None:
        rope = kw.get("rope_parameters") or {}
        if "rope_theta" not in kw and isinstance(rope, dict) and "rope_theta" in rope:
            kw["rope_theta"] = rope["rope_theta"]
        for k, default in self._fields.items():
            setattr(self, k, kw.get(k, default))
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        self._extra = {k: v for k, v in kw.items() if k not in self._fields}
        if self.num_key_value_heads != self.num_attention_heads:
            raise ValueError("grouped-query attention is not part of the reference path (MHA only)")
        if self.hidden_act != "silu" or self.attention_bias or self.mlp_bias:
            raise ValueError("only bias-free SwiGLU(silu) LLaMA blocks are supported")

    def to_dict(self) -> Dict[str, Any]:
        d = dict(self._extra)
        d.update({k: getattr(self, k) for k in self._fields})
        d["model_type"] = "llama"
        return d

    @staticmethod
    def coerce(obj: Union["NetConfig", Dict, Any, None]) -> "NetConfig":
        if obj is None:
            return NetConfig()
        if isinstance(obj, NetConfig):
            return obj
        if isinstance(obj, dict):
            return NetConfig(**obj)
        return NetConfig(**obj.to_dict())  # HF LlamaConfig


class MIDIModelConfig:
    model_type = "midi_model"

    def __init__(self, tokenizer=None, net_config=None, net_token_config=None, **kwargs: Any) -> None:
        if tokenizer:
            if isinstance(tokenizer, dict):
                tok = MIDITokenizer(tokenizer["version"])
                tok.set_optimise_midi(tokenizer["optimise_midi"])
                tokenizer = tok
            self.tokenizer = tokenizer
        else:
            self.tokenizer = MIDITokenizer()
        self.net_config = NetConfig.coerce(net_config)
        self.net_token_config = NetConfig.coerce(net_token_config)
        self.n_embd = self.net_token_config.hidden_size
        self._kwargs = {k: v for k, v in kwargs.items() if k not in ("model_type", "n_embd")}

    # ---- serialisation (HF-compatible layout) -------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        d = dict(self._kwargs)
        d["model_type"] = self.model_type
        d["n_embd"] = self.n_embd
        d["net_config"] = self.net_config.to_dict()
        d["net_token_config"] = self.net_token_config.to_dict()
        d["tokenizer"] = self.tokenizer.to_dict()
        return d

    def to_json_string(self) -> str:
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, save_directory: str) -> None:
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            f.write(self.to_json_string())

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "MIDIModelConfig":
        d = dict(d)
        return cls(d.pop("tokenizer", None), d.pop("net_config", None), d.pop("net_token_config", None), **d)

    @classmethod
    def from_json_file(cls, path: str) -> "MIDIModelConfig":
        with open(path) as f:
            return cls.from_dict(json.load(f))

    @classmethod
    def from_pretrained(cls, directory: str) -> "MIDIModelConfig":
        return cls.from_json_file(os.path.join(directory, "config.json"))

    def __str__(self) -> str:
        return json.dumps({"net": self.net_config.to_dict(), "net_token": self.net_token_config.to_dict()}, indent=4)

    # ---- presets -------------------------------------------------------------------------
    @staticmethod
    def get_config(tokenizer_ver="v2", optimise_midi=True, n_layer=12, n_head=16, n_embd=1024, n_inner=4096):
        """Event-level net (n_layer, n_head, n_inner) + a token-level net a quarter its depth,
        heads and MLP width at the same hidden size (midi_model.py:63-76)."""
        tok = MIDITokenizer(tokenizer_ver)
        tok.set_optimise_midi(optimise_midi)
        common = dict(vocab_size=tok.vocab_size, hidden_size=n_embd, pad_token_id=tok.pad_id,
                      max_position_embeddings=4096, use_cache=False)
        net = NetConfig(num_attention_heads=n_head, num_hidden_layers=n_layer, intermediate_size=n_inner, **common)
        net_token = NetConfig(num_attention_heads=n_head // 4, num_hidden_layers=n_layer // 4,
                              intermediate_size=n_inner // 4, **common)
        return MIDIModelConfig(tok, net, net_token)

    @staticmethod
    def from_name(name="tv2o-medium"):
        tv, size = name.split("-")
        tv = tv[1:]
        optimise = tv.endswith("o")
        if optimise:
            tv = tv[:-1]
        if tv not in ("v1", "v2"):
            raise ValueError(f"Unknown tokenizer version {tv}")
        depth = {"medium": 12, "large": 24}.get(size)
        if depth is None:
            raise ValueError(f"Unknown model size {size}")
        return MIDIModelConfig.get_config(tokenizer_ver=tv, optimise_midi=optimise,
                                          n_layer=depth, n_head=16, n_embd=1024, n_inner=4096)
