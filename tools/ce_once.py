#!/usr/bin/env python
"""mh_cross_entropy on the training step's chunk (32768 rows x 3406 of 3408, bf16) and two ragged shapes: time per launch (HIP
events) and a digest of the outputs, for A/B runs of two builds of the library (MH_LIB_PATH=... python tools/ce_once.py)."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for (R, V, ldl, arg) in [(32768, 3406, 3408, False), (32768, 3406, 3408, True), (4099, 2049, 2056, False), (1000, 3406, 3408, True), (777, 500, 504, False)]:
    logits = (torch.randn((R, ldl), device="cuda", generator=g) * 3).to(torch.bfloat16)
    tgt = torch.randint(0, V, (R,), device="cuda", generator=g)
    tgt[::7] = 0
    loss = torch.empty((R,), device="cuda"); dl = torch.empty_like(logits)
    am = torch.empty((R,), dtype=torch.long, device="cuda") if arg else None
    scale = torch.full((1,), 1.0 / 1234.0, device="cuda")
    run = lambda: ops.cross_entropy(logits, V, tgt, loss, dlogits=dl, scale_dev=scale, argmax_out=am)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    h = hashlib.sha256(dl.cpu().view(torch.int16).numpy().tobytes() + loss.cpu().numpy().tobytes() + (am.cpu().numpy().tobytes() if arg else b"")).hexdigest()[:16]
    ref = torch.log_softmax(logits[:, :V].float(), -1)
    ok = torch.allclose(loss, torch.where(tgt != 0, -ref.gather(1, tgt[:, None])[:, 0], torch.zeros_like(loss)), atol=2e-5, rtol=1e-5)
    print(f"R={R} V={V} ldl={ldl} argmax={arg}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  digest {h}  loss vs torch {ok}", flush=True)
