#!/bin/bash
# HBM-side traffic of the training step per kernel: two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE + L2 hits) over a
# short bench run, reduced on the GPU box to per-kernel sums (the raw counter CSVs are too large to copy back).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
n=0
for g in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  out=/tmp/pmcb_$n
  rm -rf $out
  (cd /tmp && timeout 900 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-events --no-extras > $ROOT/gpurun_out/pmcb_$n.log 2>&1)
  n=$((n+1))
done
python - <<'PY' > gpurun_out/pmc_bench_traffic.txt
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for n in (0, 1):
    for r in csv.DictReader(open(f"/tmp/pmcb_{n}/pmc_counter_collection.csv")):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
print("# per kernel over bench.py --steps 1 --warmup 1 (2 steps): dispatches, FETCH_SIZE KiB (x2 for 16-B/lane streams on gfx950), WRITE_SIZE KiB, L2 hit rate")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))[:25]:
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(f"{k:70s} n={len(cnt[k]) // 1:5d} fetch={v.get('FETCH_SIZE', 0):14.0f} write={v.get('WRITE_SIZE', 0):14.0f} hit={h / max(1.0, h + m):.3f}")
PY
python - <<'PY' > gpurun_out/pmc_gemm_traffic.json
# memory-side traffic of the dominant kernel per launch, for bench.py's roofline.traffic (provenance: this file's name)
import csv, json, collections
agg = collections.defaultdict(float); disp = set()
for n in (0, 1):
    for r in csv.DictReader(open(f"/tmp/pmcb_{n}/pmc_counter_collection.csv")):
        if "gemm_pp256_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            if n == 0:
                disp.add(r["Dispatch_Id"])
L = max(1, len(disp))
fetch_kib, write_kib = agg.get("FETCH_SIZE", 0.0) / L, agg.get("WRITE_SIZE", 0.0) / L
h, m = agg.get("TCC_HIT_sum", 0.0), agg.get("TCC_MISS_sum", 0.0)
read_b, write_b = 2.0 * fetch_kib * 1024.0, write_kib * 1024.0
print(json.dumps({
    "kernel": "gemm_pp256_kernel (all instantiations)", "launches_profiled": L,
    "bytes_per_launch": read_b + write_b, "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
    "fetch_size_kib_per_launch_raw": fetch_kib, "write_size_kib_per_launch_raw": write_kib,
    "l2_hit_rate": h / max(1.0, h + m),
    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (two separate passes, --kernel-trace only) over "
           "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-events --no-extras; FETCH_SIZE doubled for 16-B/lane streams "
           "on gfx950 (MI355X_MICROARCH.md, HBM section); memory-side = fabric requests of the L2s (Infinity-Cache hits included)"}))
PY
cat gpurun_out/pmc_bench_traffic.txt; cat gpurun_out/pmc_gemm_traffic.json
