"""Import shim: the package directory is ``midi-model_amd/`` (not a valid Python identifier), so
``import midi_model_amd`` lands here and this file swaps itself for the real package."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "midi-model_amd")
_spec = _ilu.spec_from_file_location("midi_model_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["midi_model_amd"] = _mod
_spec.loader.exec_module(_mod)
