#!/bin/bash
# PMC of the 4-wave probe's main loop against the production kernel, [32768 x 1024 x 16384]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; R=$(pwd)
export TMPDIR=/tmp
for v in 0 3; do
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $grp | cut -c1-12 | tr ' ' '_')
    rm -rf /tmp/p4_$v_$tag
    (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/p4_${v}_$tag -o p -- $R/tools/bin/gemm4w_probe 32768 1024 16384 $v > /dev/null 2>&1)
    python3 - /tmp/p4_${v}_$tag $v >> $O/r05_probe_pmc.txt <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in agg.items():
    if "gemm" in k: print("variant", sys.argv[2], k, {c: round(v / 5) for c, v in d.items()})
PY
  done
done
cat $O/r05_probe_pmc.txt
