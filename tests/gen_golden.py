"""Generate tests/golden/* by running the REAL reference (``/root/reference/midi_model.py``) on CPU
in fp32.  Runs only in the build container (the GPU box has no /root/reference); the outputs are
committed.  Usage:  python tests/gen_golden.py

The reference imports ``peft`` (midi_model.py:9) which is not installed; the names are only used by
``load_merge_lora`` (:109-114), so a stub module is placed on sys.path.  ``train.py`` cannot be
imported (needs ``lightning``): its step (train.py:168-188), optimiser (:121-151) and the Trainer's
``gradient_clip_val=1.0`` / LambdaLR wiring are driven here with the real torch objects.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "golden")
REF = "/root/reference"


def import_reference():
    stub = tempfile.mkdtemp(prefix="peft_stub_")
    os.makedirs(os.path.join(stub, "peft"))
    with open(os.path.join(stub, "peft", "__init__.py"), "w") as f:
        f.write("PeftConfig = LoraModel = LoraConfig = TaskType = None\n"
                "def load_peft_weights(*a, **k): raise NotImplementedError\n"
                "def set_peft_model_state_dict(*a, **k): raise NotImplementedError\n")
    # the reference must win over the repo's own drop-in ``midi_model.py``
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    sys.path.insert(0, REF)
    sys.path.insert(0, stub)
    import midi_model as ref_model  # noqa
    import midi_tokenizer as ref_tok  # noqa
    assert ref_model.__file__.startswith(REF)
    return ref_model, ref_tok


def load_oracle():
    import importlib.util
    spec = importlib.util.spec_from_file_location("midi_oracle", os.path.join(ROOT, "oracle", "midi_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def tokenizer_tables(tok):
    d = tok.to_dict()
    d["event_ids"] = tok.event_ids
    d["parameter_ranges"] = {k: [v[0], v[-1]] for k, v in tok.parameter_ids.items()}
    d["parameter_order"] = list(tok.parameter_ids.keys())
    return d


def build_ref(ref_model, shp, sd):
    cfg = ref_model.MIDIModelConfig.get_config("v2", True, shp.n_layer, shp.n_head, shp.n_embd, shp.n_inner)
    model = ref_model.MIDIModel(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    model.eval()
    return model


def ref_train_loss(model, batch):
    """train.py:168-188 driven verbatim on the real model."""
    import torch.nn.functional as F
    x = batch[:, :-1].contiguous()
    y = batch[:, 1:].contiguous()
    hidden = model.forward(x)
    hidden = hidden.reshape(-1, hidden.shape[-1])
    y = y.reshape(-1, y.shape[-1])
    xt = y[:, :-1]
    logits = model.forward_token(hidden, xt)
    loss = F.cross_entropy(logits.view(-1, model.tokenizer.vocab_size), y.view(-1), reduction="mean",
                           ignore_index=model.tokenizer.pad_id)
    return loss, logits, hidden


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_model, ref_tok = import_reference()
    orc = load_oracle()
    torch.set_num_threads(os.cpu_count())

    # ---- 1. tokenizer tables -------------------------------------------------------------
    for ver in ("v1", "v2"):
        with open(os.path.join(OUT, f"tokenizer_{ver}.json"), "w") as f:
            json.dump(tokenizer_tables(ref_tok.MIDITokenizer(ver)), f, indent=1, sort_keys=True)

    tok = ref_tok.MIDITokenizer("v2")
    from transformers import DynamicCache

    # ---- 2. tiny model: forward, cache, loss, grads, optimiser, generate ------------------------
    shp = orc.Shape(n_layer=4, n_head=4, n_embd=256, n_inner=512, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=1)
    model = build_ref(ref_model, shp, sd)
    batch = orc.synthetic_events(tok, 2, 17, seed=2)
    batch[1, 14:] = tok.pad_id  # a ragged row: trailing pad events (collate padding, train.py:86-90)
    g = {}
    with torch.no_grad():
        x = batch[:, :-1]
        hid = model.forward(x)
        g["hidden"] = hid.numpy()
        c = DynamicCache()
        h_a = model.forward(x[:, :11], cache=c)
        h_b = model.forward(x[:, 11:], cache=c)
        g["hidden_cached"] = torch.cat([h_a, h_b], 1).numpy()
    for p in model.parameters():
        p.grad = None
    loss, logits, _ = ref_train_loss(model, batch)
    loss.backward()
    g["loss"] = np.float64(loss.item())
    g["logits_sub"] = logits.detach()[:, :, ::16].numpy()
    g["logits_lse"] = torch.logsumexp(logits.detach(), -1).numpy()
    g["logits_argmax"] = logits.detach().argmax(-1).numpy()
    names = [n for n, _ in model.named_parameters()]
    g["grad_names"] = np.array(names)
    g["grad_norms"] = np.array([p.grad.norm().item() for p in model.parameters()], dtype=np.float64)
    named = dict(model.named_parameters())
    for key in ("net.layers.0.input_layernorm.weight", "net.norm.weight", "net_token.layers.0.post_attention_layernorm.weight"):
        g["grad:" + key] = named[key].grad.numpy()
    for key in ("net.layers.1.self_attn.q_proj.weight", "net.layers.3.mlp.down_proj.weight",
                "net_token.layers.0.self_attn.v_proj.weight", "net_token.layers.0.mlp.gate_proj.weight",
                "lm_head.weight", "net.embed_tokens.weight", "net_token.embed_tokens.weight"):
        g["grad:" + key] = named[key].grad[:64:3, ::5].numpy()
    # accuracy (train.py:153-166 restated with the same torch ops)
    y = batch[:, 1:].reshape(-1, 8)
    out = logits.detach().argmax(-1).flatten()
    lab = y.flatten()
    keep = lab != tok.pad_id
    g["acc"] = np.float64(((out[keep] == lab[keep]).float().sum() / keep.sum()).item())

    # optimiser: 3 steps of clip(1.0) + AdamW + LambdaLR(warmup=2, max_step=10), lr 1e-2 (large, so
    # the update is visible in fp32), acc_grad = 1
    params = list(model.named_parameters())
    no_decay = ["bias", "norm"]
    groups = [{"params": [p for n, p in params if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in params if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: orc.lr_lambda(s, 2, 10))
    losses, gnorms, lrs = [], [], []
    for step in range(3):
        b = orc.synthetic_events(tok, 2, 17, seed=10 + step)
        opt.zero_grad(set_to_none=True)
        l, _, _ = ref_train_loss(model, b)
        l.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        lrs.append(sched.get_last_lr()[0])
        opt.step()
        sched.step()
        losses.append(l.item())
        gnorms.append(gn.item())
    g["opt_losses"] = np.array(losses, dtype=np.float64)
    g["opt_gnorms"] = np.array(gnorms, dtype=np.float64)
    g["opt_lrs"] = np.array(lrs, dtype=np.float64)
    named = dict(model.named_parameters())
    g["opt_param_norms"] = np.array([named[n].detach().norm().item() for n in names], dtype=np.float64)
    for key in ("net.layers.0.input_layernorm.weight", "net.norm.weight"):
        g["opt:" + key] = named[key].detach().numpy().copy()
    g["opt:lm_head.weight"] = named["lm_head.weight"].detach()[:64:3, ::5].numpy().copy()
    g["opt:net.layers.2.mlp.up_proj.weight"] = named["net.layers.2.mlp.up_proj.weight"].detach()[:64:3, ::5].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "tiny_train.npz"), **g)

    # generation (fresh weights): sampled with a seeded generator, greedy, and with a prompt
    model = build_ref(ref_model, shp, sd)
    gg = {}
    gen = torch.Generator().manual_seed(1234)
    gg["sampled_b3"] = model.generate(None, batch_size=3, max_len=14, temp=1.0, top_p=0.98, top_k=20, generator=gen)
    gg["greedy_b2"] = model.generate(None, batch_size=2, max_len=14, temp=1.0, top_p=0.98, top_k=1,
                                     generator=torch.Generator().manual_seed(0))
    prompt = orc.synthetic_events(tok, 1, 6, seed=5)[0].numpy()
    gg["prompt"] = prompt
    gg["prompt_b2"] = model.generate(prompt, batch_size=2, max_len=12, temp=0.9, top_p=0.9, top_k=8,
                                     generator=torch.Generator().manual_seed(77))
    # sampler on fixed inputs
    pr = torch.softmax(3.0 * torch.randn((4, 1, tok.vocab_size), generator=torch.Generator().manual_seed(3)), -1)
    gg["sampler_probs_seed"] = np.int64(3)
    gg["sampler_out"] = model.sample_top_p_k(pr.clone(), 0.9, 12, generator=torch.Generator().manual_seed(9)).numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_generate.npz"), **gg)

    # ---- 3. tv2o-medium (the real shape: head_dim 64 / 256) ------------------------------------
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    cfg = ref_model.MIDIModelConfig.from_name("tv2o-medium")
    model = ref_model.MIDIModel(cfg)
    model.load_state_dict(sd, strict=True)
    model.eval()
    batch = orc.synthetic_events(tok, 1, 33, seed=4)
    m = {}
    with torch.no_grad():
        loss, logits, hidden = ref_train_loss(model, batch)
    m["loss"] = np.float64(loss.item())
    m["hidden_sub"] = hidden[:, ::4].numpy()
    m["logits_sub"] = logits[:, :, ::32].numpy()
    m["logits_lse"] = torch.logsumexp(logits, -1).numpy()
    m["logits_argmax"] = logits.argmax(-1).numpy()
    top2 = logits.topk(2, -1).values
    m["logits_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    m["n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    m["state_dict_keys"] = np.array(list(model.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "medium_forward.npz"), **m)
    print("golden written to", OUT, {k: os.path.getsize(os.path.join(OUT, k)) for k in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
