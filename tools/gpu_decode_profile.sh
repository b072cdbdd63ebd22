#!/bin/bash
# Evidence for bench.py --mode generate (BASELINE configs[3]: batch 64, 1024 new events): (1) rocprofv3 kernel trace -> per-kernel
# time per generated event; (2) two PMC passes (FETCH_SIZE; WRITE_SIZE + L2 hits) -> memory-side bytes per event step, the
# `traffic` of the generate roofline object.  Usage: tools/gpu_decode_profile.sh <tag> [events=1024]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out; T=${1:-dec}; EV=${2:-1024}
CMD="python $R/bench.py --mode generate --gen-events $EV --steps 1 --warmup 1 --no-cpu-baseline"
(cd /tmp && rm -rf /tmp/dtr_$T && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr_$T -o gen -- $CMD > $O/${T}_trace_bench.log 2>&1)
F=$(find /tmp/dtr_$T -name "*kernel_trace.csv" | head -1)
python - "$F" $EV > $O/${T}_generate_kernels_per_event.txt <<'PY'
import csv, sys, re, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
ev = 2 * int(sys.argv[2])  # warmup call + timed call
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:90]
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    tot[short(n)][0] += e - s; tot[short(n)][1] += 1
busy = sum(e - s for s, e, _ in rows); span = rows[-1][1] - rows[0][0]
print(f"# rocprofv3 --kernel-trace of bench.py --mode generate, {ev} generated events (batch 64) in two generate() calls")
print(f"# kernels {len(rows)} = {len(rows) / ev:.1f} per event; busy {busy / ev / 1e3:.1f} us per event; trace span {span / ev / 1e3:.1f} us per event")
print(f"# {'kernel':90s} launches/event   us/launch   us/event")
for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {n:90s} {c / ev:8.2f} {t / c / 1e3:11.2f} {t / ev / 1e3:10.1f}")
PY
n=0
for g in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/dpmc_$n
  (cd /tmp && timeout 1500 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/dpmc_$n -o pmc -- $CMD > $O/${T}_pmc_$n.log 2>&1)
  n=$((n+1))
done
python - $EV > $O/${T}_pmc_generate_traffic.json <<'PY'
import csv, json, collections, sys
ev = 2 * int(sys.argv[1])
agg = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float))
for n in (0, 1):
    for r in csv.DictReader(open(f"/tmp/dpmc_{n}/pmc_counter_collection.csv")):
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
read_b, write_b = 2.0 * agg["FETCH_SIZE"] * 1024.0 / ev, agg["WRITE_SIZE"] * 1024.0 / ev
h, m = agg["TCC_HIT_sum"], agg["TCC_MISS_sum"]
top = sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))[:8]
print(json.dumps({
    "what": "memory-side bytes per generated event (one net step + the token steps), all kernels of bench.py --mode generate",
    "events_profiled": ev, "bytes_per_event_step": read_b + write_b, "read_bytes_per_event_step": read_b, "write_bytes_per_event_step": write_b,
    "l2_hit_rate": h / max(1.0, h + m),
    "top_readers_bytes_per_event": {k: 2.0 * v.get("FETCH_SIZE", 0) * 1024.0 / ev for k, v in top},
    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (two separate passes, --kernel-trace only) over "
           "bench.py --mode generate --steps 1 --warmup 1 (both calls counted); FETCH_SIZE doubled for 16-B/lane streams on gfx950 "
           "(MI355X_MICROARCH.md, HBM section); memory-side = fabric requests of the L2s (Infinity-Cache hits included)"}))
PY
cat $O/${T}_generate_kernels_per_event.txt; cat $O/${T}_pmc_generate_traffic.json; tail -2 $O/${T}_trace_bench.log | cut -c1-300
