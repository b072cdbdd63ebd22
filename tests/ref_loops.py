"""TEST FIXTURE (category b: the reference's own caller code, kept verbatim on purpose).

``serving_loop`` is the body of the reference's ``generate`` in app.py:27-120 with its three module globals
(``model``, ``tokenizer``, transformers' ``DynamicCache``) turned into arguments and the tqdm bar dropped -- nothing else
is changed, because the point of the test that uses it is that the reference's OWN loop (a real
``transformers.DynamicCache`` created by the caller, ``model.forward(x, cache=...)``, ``model.forward_token(hidden, x,
cache=...)``, ``model.sample_top_p_k``) drives our ``MIDIModel`` unchanged.  It is never imported by the package.
"""
import numpy as np
import torch
import torch.nn.functional as F


@torch.inference_mode()
def serving_loop(model, tokenizer, DynamicCache, prompt=None, batch_size=1, max_len=512, temp=1.0, top_p=0.98, top_k=20,
                 disable_patch_change=False, disable_control_change=False, disable_channels=None, generator=None):
    if disable_channels is not None:
        disable_channels = [tokenizer.parameter_ids["channel"][c] for c in disable_channels]
    else:
        disable_channels = []
    max_token_seq = tokenizer.max_token_seq
    if prompt is None:
        input_tensor = torch.full((1, max_token_seq), tokenizer.pad_id, dtype=torch.long, device=model.device)
        input_tensor[0, 0] = tokenizer.bos_id  # bos
        input_tensor = input_tensor.unsqueeze(0)
        input_tensor = torch.cat([input_tensor] * batch_size, dim=0)
    else:
        if len(prompt.shape) == 2:
            prompt = prompt[None, :]
            prompt = np.repeat(prompt, repeats=batch_size, axis=0)
        elif prompt.shape[0] == 1:
            prompt = np.repeat(prompt, repeats=batch_size, axis=0)
        elif len(prompt.shape) != 3 or prompt.shape[0] != batch_size:
            raise ValueError(f"invalid shape for prompt, {prompt.shape}")
        prompt = prompt[..., :max_token_seq]
        if prompt.shape[-1] < max_token_seq:
            prompt = np.pad(prompt, ((0, 0), (0, 0), (0, max_token_seq - prompt.shape[-1])),
                            mode="constant", constant_values=tokenizer.pad_id)
        input_tensor = torch.from_numpy(prompt).to(dtype=torch.long, device=model.device)
    input_tensor = input_tensor[:, -4096:]
    cur_len = input_tensor.shape[1]
    cache1 = DynamicCache()
    past_len = 0
    while cur_len < max_len:
        end = [False] * batch_size
        hidden = model.forward(input_tensor[:, past_len:], cache=cache1)[:, -1]
        next_token_seq = None
        event_names = [""] * batch_size
        cache2 = DynamicCache()
        for i in range(max_token_seq):
            mask = torch.zeros((batch_size, tokenizer.vocab_size), dtype=torch.int64, device=model.device)
            for b in range(batch_size):
                if end[b]:
                    mask[b, tokenizer.pad_id] = 1
                    continue
                if i == 0:
                    mask_ids = list(tokenizer.event_ids.values()) + [tokenizer.eos_id]
                    if disable_patch_change:
                        mask_ids.remove(tokenizer.event_ids["patch_change"])
                    if disable_control_change:
                        mask_ids.remove(tokenizer.event_ids["control_change"])
                    mask[b, mask_ids] = 1
                else:
                    param_names = tokenizer.events[event_names[b]]
                    if i > len(param_names):
                        mask[b, tokenizer.pad_id] = 1
                        continue
                    param_name = param_names[i - 1]
                    mask_ids = tokenizer.parameter_ids[param_name]
                    if param_name == "channel":
                        mask_ids = [i for i in mask_ids if i not in disable_channels]
                    mask[b, mask_ids] = 1
            mask = mask.unsqueeze(1)
            x = next_token_seq
            if i != 0:
                hidden = None
                x = x[:, -1:]
            logits = model.forward_token(hidden, x, cache=cache2)[:, -1:]
            scores = torch.softmax(logits / temp, dim=-1) * mask
            samples = model.sample_top_p_k(scores, top_p, top_k, generator=generator)
            if i == 0:
                next_token_seq = samples
                for b in range(batch_size):
                    if end[b]:
                        continue
                    eid = samples[b].item()
                    if eid == tokenizer.eos_id:
                        end[b] = True
                    else:
                        event_names[b] = tokenizer.id_events[eid]
            else:
                next_token_seq = torch.cat([next_token_seq, samples], dim=1)
                if all([len(tokenizer.events[event_names[b]]) == i for b in range(batch_size) if not end[b]]):
                    break
        if next_token_seq.shape[1] < max_token_seq:
            next_token_seq = F.pad(next_token_seq, (0, max_token_seq - next_token_seq.shape[1]),
                                   "constant", value=tokenizer.pad_id)
        next_token_seq = next_token_seq.unsqueeze(1)
        input_tensor = torch.cat([input_tensor, next_token_seq], dim=1)
        past_len = cur_len
        cur_len += 1
        yield next_token_seq[:, 0].cpu().numpy()
        if all(end):
            break
