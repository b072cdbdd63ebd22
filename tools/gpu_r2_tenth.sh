#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py::test_attention_fwd_bwd -q -m gpu --tb=short -p no:cacheprovider -x > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
for cfg in "1 2" "2 2" "2 1" "1 2" "2 2"; do
  set -- $cfg
  MH_ATTN_FWD=1 MH_ATTN_FWD_QB=$1 MH_ATTN_FWD_WPS=$2 timeout 300 python bench.py --mode block --steps 10 --warmup 3 > $O/block_q$1_w$2.log 2> $O/block_q$1_w$2.err
  python - <<PY
import json
d=json.loads(open("$O/block_q$1_w$2.log").read().strip().splitlines()[-1])
b=d["block"]; print("qb $1 wps $2:", "ms", round(b["ms_per_block"],3), "frac", round(b["roofline"]["frac"],4), "attn TF", round(b["attention_tflops"],1), {k:round(v["us_per_call"],1) for k,v in b["kernels"].items() if "attn" in k})
PY
done
for i in 1 2; do timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generate', round(d['value']), 'ev/s', round(d['config']['ms_per_event_step'],4), 'ms/event frac', round(d['roofline']['frac'],4))"; done
for ev in "" "--no-gemm-events"; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras $ev 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train $ev', round(d['value']), 'ev/s', round(d['ms_per_step'],2))"; done
tail -n 4 $O/retest.log
