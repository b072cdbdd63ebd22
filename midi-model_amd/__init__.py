"""MI355X-native hot path of SkyTNT/midi-model: the two-level LLaMA stack (MIDIModel.net over
events, MIDIModel.net_token over each event's token octet), its data-parallel training step and
the KV-cached generate loop, on hand-written gfx950 HIP kernels behind a C-ABI (include/midihip.h).

Importing the package is cheap and GPU-free; the HIP library is loaded on first kernel use and
its absence is a hard error (there is no CPU fallback)."""
from .config import MIDIModelConfig, NetConfig, config_name_list  # noqa: F401
from .tokenizer import MIDITokenizer, MIDITokenizerV1, MIDITokenizerV2  # noqa: F401

__all__ = ["MIDIModelConfig", "NetConfig", "config_name_list", "MIDITokenizer", "MIDITokenizerV1",
           "MIDITokenizerV2", "MIDIModel"]


def __getattr__(name):
    if name == "MIDIModel":
        from .model import MIDIModel
        return MIDIModel
    raise AttributeError(name)
