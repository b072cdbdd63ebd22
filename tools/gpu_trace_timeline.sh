#!/bin/bash
# kernel timeline of one training step (rocprofv3 --kernel-trace): per launch its start offset, duration and the idle gap
# in front of it -- shows what a HIP-event window around a launch really contains.  Usage: tools/gpu_trace_timeline.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$(pwd)/gpurun_out
T=${1:-trace}
R=$(pwd)
(cd /tmp && rm -rf /tmp/tr_$T && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$T -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/${T}_bench.log 2>&1)
F=$(find /tmp/tr_$T -name "*kernel_trace.csv" | head -1)
python - "$F" > $O/${T}_timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"gemm_pp256_kernel<[^>]*>", n)
    if m: return m.group(0)
    n = re.sub(r"\(.*", "", n)
    return n[-60:]
# last step only: find the last adamw launch group; print the ~1500 launches before it
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    out.append((s - t0, e - s, gap, short(r["Kernel_Name"])))
    prev_end = max(prev_end or 0, e)
tail = out[-1400:]
for s, d, g, n in tail:
    print(f"{s/1e3:12.1f} us  dur {d/1e3:9.1f}  gap {g/1e3:8.1f}  {n}")
tot_gap = sum(g for _, _, g, _ in tail if g > 0)
print(f"# sum of idle gaps over these {len(tail)} launches: {tot_gap/1e3:.1f} us; span {(tail[-1][0]+tail[-1][1]-tail[0][0])/1e3:.1f} us")
PY
grep -n "pp256_kernel<false, false, 0, 3>" $O/${T}_timeline.txt | head -3
tail -1 $O/${T}_timeline.txt
