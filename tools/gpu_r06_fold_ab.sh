#!/bin/bash
# r06: the training fold of the RMSNorms -- kernel tests, the model-level parity tests it touches, and the training step with and
# without it on one box (MH_NORM_FOLD_TRAIN=0 = the r01-r05 schedule)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06_fold}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "row_scale" 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_parity_long_gpu.py -q -x -k "training or step or lean or accumulation or lora" 2>&1 | tail -15 > gpurun_out/${T}_model_tests.txt; tail -6 gpurun_out/${T}_model_tests.txt
for i in 1 2; do
  MH_NORM_FOLD_TRAIN=0 python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfolded', round(d['value']), round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernel_families'].items()})"
  python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('folded  ', round(d['value']), round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernel_families'].items()})"
done 2>&1 | tee gpurun_out/${T}_ab.txt
