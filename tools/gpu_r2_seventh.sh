#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py::test_attention_fwd_bwd tests/test_kernels_gpu.py::test_attention_mfma_vs_plain_on_device "tests/test_parity_long_gpu.py::test_flash_attention_at_benchmarked_length" "tests/test_parity_long_gpu.py::test_swiglu_epilogues_at_benchmarked_shape" "tests/test_parity_long_gpu.py::test_forward_at_benchmarked_length" -q -m gpu --tb=short -p no:cacheprovider > $O/retest.log 2>&1
echo "retest rc=$?" >> $O/retest.log
for cfg in "1 2" "2 2" "2 3" "1 2" "2 2"; do
  set -- $cfg
  MH_ATTN_FWD=$1 MH_ATTN_FWD_WPS=$2 timeout 300 python bench.py --mode block --steps 10 --warmup 3 > $O/block_f$1_w$2.log 2> $O/block_f$1_w$2.err
  python - <<PY
import json
d=json.loads(open("$O/block_f$1_w$2.log").read().strip().splitlines()[-1])
b=d["block"]; print("form $1 wps $2:", "ms", round(b["ms_per_block"],3), "frac", round(b["roofline"]["frac"],4), "attn TF", round(b["attention_tflops"],1), {k:round(v["us_per_call"],1) for k,v in b["kernels"].items()})
PY
done
MH_ATTN_FWD=2 timeout 300 python bench.py --mode block --block-save --steps 10 --warmup 3 > $O/block_save.log 2>/dev/null
python -c "
import json
d=json.loads(open('$O/block_save.log').read().strip().splitlines()[-1]); b=d['block']; print('training-forward form:', round(b['ms_per_block'],3), 'ms frac', round(b['roofline']['frac'],4))"
timeout 300 python tools/decode_probe.py 1b > $O/decode_probe_1b.txt 2>&1
tail -n 4 $O/retest.log; cat $O/decode_probe_1b.txt
