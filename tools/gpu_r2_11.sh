#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do for spec in 0 1; do MH_DECODE_SPEC=$spec timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec=$spec generate', round(d['value']), 'ev/s', round(d['config']['ms_per_event_step'],4), 'ms/event frac', round(d['roofline']['frac'],4))"; done; done
