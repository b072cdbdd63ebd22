#!/bin/bash
# Matrix-pipe utilisation per kernel over the training step (north_star: "evidenced by rocprof ... MFMA-busy counters"):
# one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# SQ_INSTS_MFMA SQ_INSTS_VALU) + one (GRBM_GUI_ACTIVE), --kernel-trace only, over bench.py --steps 1 --warmup 1.
#   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)   (MFMA_BUSY counts cycles: 16 per
#   16x16x32 bf16 MFMA, 32 per 32x32x16; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
# Usage: tools/gpu_pmc_mfma.sh <tag>   -> gpurun_out/<tag>_pmc_mfma_busy_by_kernel.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd); T=${1:-mfma}
n=0
for g in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
  out=/tmp/pmcm_$n
  rm -rf $out
  (cd /tmp && timeout 900 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-events --no-extras > $ROOT/gpurun_out/pmcm_$n.log 2>&1)
  n=$((n+1))
done
python - <<'PY' > gpurun_out/${T}_pmc_mfma_busy_by_kernel.txt
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
import glob
wall = collections.defaultdict(float)   # kernel -> summed wall time (ns) of its dispatches IN THE GRBM PASS (pass 1)
for n in (0, 1):
    seen = set()
    for r in csv.DictReader(open(f"/tmp/pmcm_{n}/pmc_counter_collection.csv")):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        if n == 1 and r["Dispatch_Id"] not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            seen.add(r["Dispatch_Id"]); wall[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if n == 1 and not wall:   # (older layout: the timestamps live in the kernel trace of the same pass)
        for f in glob.glob("/tmp/pmcm_1/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
                wall[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print("# per kernel over bench.py --steps 1 --warmup 1 (2 training steps, tv2o-medium 16 x 2048 bf16): matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /")
print("# (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); VALU per MFMA = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA; waves: parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES,")
print("# issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, issuing = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES.  Two separate --pmc passes (--kernel-trace only).")
print("# clock GHz = GRBM_GUI_ACTIVE / 8 XCDs / the kernel's wall time in the same (profiled) pass: the clock the part HOLDS under that kernel (nominal 2.4);")
print("# 'busy at nominal' = mfma busy x clock / 2.4 = the fraction of the 2.5 PFLOP/s peak's cycles the matrix pipe worked.")
print(f"# {'kernel':64s} {'launches':>8s} {'mfma busy':>9s} {'valu/mfma':>9s} {'parked':>7s} {'stalled':>7s} {'issuing':>7s} {'clock GHz':>9s} {'busy@2.4':>8s}")
rows = []
for k, v in agg.items():
    g = v.get("GRBM_GUI_ACTIVE", 0.0)
    if g <= 0 or v.get("SQ_INSTS_MFMA", 0) <= 0: continue
    simd_cycles = g / 8.0 * 1024.0
    wc = max(1.0, v.get("SQ_WAVE_CYCLES", 0))
    ghz = (g / 8.0) / wall[k] if wall.get(k) else float("nan")
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    rows.append((g, k, len(cnt[k]) // 2, busy, (v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"],
                 v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc, ghz, busy * ghz / 2.4))
for g, k, n, busy, vpm, a, b, c, ghz, bn in sorted(rows, reverse=True):
    print(f"  {k:64s} {n:8d} {busy:9.3f} {vpm:9.2f} {a:7.3f} {b:7.3f} {c:7.3f} {ghz:9.2f} {bn:8.3f}")
PY
cat gpurun_out/${T}_pmc_mfma_busy_by_kernel.txt
