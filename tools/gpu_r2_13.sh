#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for cfg in "0 0" "1 0" "1 1" "0 0" "1 1"; do set -- $cfg; MH_DECODE_SPEC=$1 MH_DECODE_COPY_PAGEABLE=$2 timeout 600 python bench.py --mode generate --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec=$1 pageable=$2 generate', round(d['value']), 'ev/s', round(d['config']['ms_per_event_step'],4), 'ms/event frac', round(d['roofline']['frac'],4))"; done
