import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_oracle():
    """The CPU oracle is test infrastructure: it is loaded by path, never importable as product code."""
    name = "midi_oracle"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", "midi_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def orc():
    return load_oracle()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def _load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return _load


TRAINED_SHAPE = dict(n_layer=4, n_head=4, n_embd=256, n_inner=1024)   # tests/gen_golden_trained.py: SHAPE


def trained_config():
    import midi_model_amd as mm
    s = TRAINED_SHAPE
    return mm.MIDIModelConfig.get_config("v2", True, s["n_layer"], s["n_head"], s["n_embd"], s["n_inner"])


def load_trained(orc):
    """tests/golden/tiny_trained.npz (tests/gen_golden_trained.py): the tiny model TRAINED with the real reference on a structured
    corpus -- peaked next-token distributions -- with weights snapped to values fp32 / bf16 / fp16 all hold exactly.
    Returns (shape, fp32 state dict, the npz)."""
    import numpy as np
    import torch
    g = np.load(os.path.join(GOLDEN, "tiny_trained.npz"), allow_pickle=False)
    sd = {}
    for k in g.files:
        if k.startswith("q8:"):
            sd[k[3:]] = torch.from_numpy(g[k].astype(np.float32)) * 2.0 ** int(g["e:" + k[3:]])
        elif k.startswith("b16:"):
            sd[k[4:]] = torch.from_numpy(g[k].copy()).view(torch.bfloat16).float()
    shp = orc.Shape(vocab=3406, **TRAINED_SHAPE)
    return shp, sd, g


@pytest.fixture(scope="session")
def trained(orc):
    return load_trained(orc)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # a gpu-marked test on a box without a GPU is an error in the run command, not a pass
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
