#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace CSV (too large to copy back whole): busy time vs wall span of the last
fraction of the run, gap histogram, per-kernel totals.  Usage: trace_summary.py <kernel_trace.csv> [tail_fraction]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.1f}%)  "
      f"avg dur {busy / len(rows) / 1e3:.2f} us  avg gap {sum(gaps) / len(gaps) / 1e3:.2f} us")
for lo, hi in ((-10**12, 0), (0, 1000), (1000, 2000), (2000, 5000), (5000, 10000), (10000, 50000), (50000, 10**12)):
    sel = [g for g in gaps if lo <= g < hi]
    print(f"  gaps in [{lo / 1e3:g}, {hi / 1e3:g}) us: {len(sel):7d}  total {sum(sel) / 1e6:8.2f} ms")
tot = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    tot[n][0] += e - s
    tot[n][1] += 1
for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {t / 1e6:8.2f} ms {c:7d} x {t / c / 1e3:7.2f} us  {n[:100]}")
