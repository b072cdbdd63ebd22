"""Drop-in for the reference module name: ``from midi_model import MIDIModel, MIDIModelConfig, config_name_list``
(app.py:20, train.py:21, export.py:8, push_to_hub.py:6) resolves to the MI355X/HIP implementation."""
from midi_model_amd import MIDIModelConfig, config_name_list  # noqa: F401
from midi_model_amd.model import MIDIModel  # noqa: F401
from midi_model_amd.tokenizer import MIDITokenizer, MIDITokenizerV1, MIDITokenizerV2  # noqa: F401
