#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for i in 1 2; do for inl in 0 1; do MH_DECODE_SPEC=0 MH_DECODE_NOISE_INLINE=$inl timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noise inline=$inl generate', round(d['value']), 'ev/s', round(d['config']['ms_per_event_step'],4), 'ms/event frac', round(d['roofline']['frac'],4))"; done; done
