#!/bin/bash
# Counter passes over ONE GEMM shape (tools/gemm_p256_probe.py): each --pmc set in its own rocprofv3 run with --kernel-trace
# only (MI355X_MICROARCH.md: PMC slots; gpurun refuses --pmc together with the API trace domains).
# Usage (through gpurun): M=12800 N=3072 K=768 EPI=8 bash tools/profile_gemm_pmc.sh <tag>
set -u
TAG=${1:-p256}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gemm_pmc_$TAG
mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- python3 tools/gemm_p256_probe.py > $OUT/p$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "gemm256" not in r["Kernel_Name"] and "gemm16" not in r["Kernel_Name"] and "copy" not in r["Kernel_Name"].lower():
            continue
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        print("%s %-32s %16.0f per dispatch (%d)" % (kn, cn, v / n, n))
PY
