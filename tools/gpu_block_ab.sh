#!/bin/bash
# same-box A/B of two settings of an environment variable on the block benchmark (bench.py --mode block), interleaved.
# usage: tools/gpu_block_ab.sh VAR A B [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do
  for v in $A $B; do
    env $VAR=$v timeout 300 python bench.py --mode block 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d.get('block',d)
print('$VAR=$v', round(b['ms_per_block'],3), 'ms frac', round(b['roofline']['frac'],4), {k:round(x['us_per_call'],1) for k,x in b['kernels'].items()})"
  done
done | tee gpurun_out/block_ab.txt
