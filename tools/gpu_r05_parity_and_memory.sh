#!/bin/bash
# r05 run 1: the trained-weights tests, the all-64-rows decode test at depth 1000, lean activations (bit identity + configs[4] peak memory)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_trained_gpu.py -q -x --tb=short -p no:cacheprovider -s > $O/r05_parity_trained.log 2>&1; echo "trained rc=$?" >> $O/r05_parity_trained.log; tail -5 $O/r05_parity_trained.log
timeout 900 python -m pytest tests/test_decode_gpu.py -q -x --tb=short -p no:cacheprovider -s -k "all_64_rows" > $O/r05_parity_depth64.log 2>&1; echo "depth64 rc=$?" >> $O/r05_parity_depth64.log; tail -4 $O/r05_parity_depth64.log
timeout 600 python - > $O/r05_parity_lean.log 2>&1 <<'PY'
import torch, midi_model_amd as mm
from midi_model_amd.train import TrainMIDIModel
from midi_model_amd.data import synthetic_events
cfg = mm.MIDIModelConfig.get_config("v2", True, 4, 4, 256, 1024)
outs=[]
for lean in (False, True):
    torch.manual_seed(0)
    m = TrainMIDIModel(cfg, accumulate_grad_batches=1).to("cuda", torch.bfloat16)
    m.lean_activations = lean
    b = synthetic_events(m.tokenizer, 4, 129, seed=3, device="cuda")
    loss = m.training_step(b)
    outs.append((loss.float().cpu(), m.grad_buffer().float().cpu().clone()))
print("lean bit-identical:", torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), outs[0][0].item())
PY
tail -2 $O/r05_parity_lean.log
timeout 900 python bench.py --mode large --steps 3 --warmup 1 > $O/r05_parity_large.json 2> $O/r05_parity_large.err; echo "large rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_parity_large.json").read().strip().splitlines()[-1])
for k in ("large","large_2x_hidden"):
    x=d[k]; print(k, round(x["value"]), "ev/s", round(x["ms_per_step"],1), "ms peak", round(x["hbm_peak_gb"],1), "headroom", round(x["hbm_headroom_gb"],1), "lean", x["lean_activations"], "batch", x["batch"])
PY
