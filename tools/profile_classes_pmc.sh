#!/bin/bash
# Matrix-pipe / wait / memory-issue counters per kernel CLASS of the round's hot paths, one pass each over
#   (a) the bench step at its defaults (ViT-B/32 bs 256, bf16: the four GEMM classes, attention, the score GEMM of the step's query) and
#   (b) one ViT-L/14 bs-128 encode in fp8 (gemm8_kernel) — tools/l14_fp8_bench.py,
# each rocprofv3 run with --kernel-trace + --pmc only (gpurun refuses --pmc next to the API trace domains).
# Usage (through gpurun): bash tools/profile_classes_pmc.sh r03   ->  gpurun_out/classes_pmc_r03/summary.csv
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/classes_pmc_$TAG
mkdir -p $OUT
SET1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
# (SQ_INST_CYCLES_VMEM is not a gfx950 counter: rocprofv3 reports it "Missing"; the family has _RD and _WR halves here:)
SET2="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
SHORT="--steps 5 --warmup 2 --retrieval-rows 0 --no-cpu-baseline --no-kernel-events --no-text"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*VMEM[A-Z_]*" | sort -u | tr "\n" " " > $OUT/vmem_counters_available.txt
i=0
for SET in "$SET1" "$SET2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/bench/p$i -o pmc -- python3 bench.py $SHORT > $OUT/bench_p$i.log 2>&1
  echo "bench pass $i rc=$?"
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/fp8/p$i -o pmc -- python3 tools/l14_fp8_bench.py > $OUT/fp8_p$i.log 2>&1
  echo "fp8 pass $i rc=$?"
  # round 6: ViT-B/32 bs 256 under the fp8 setting (K = 768 on the persistent kernel, the 50-key attention writing MXFP8)
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/b32fp8/p$i -o pmc -- python3 tools/b32_kernel_table.py > $OUT/b32fp8_p$i.log 2>&1
  echo "b32 fp8 pass $i rc=$?"
done
grep -h "Missing" $OUT/*.log | sed 's/.*Missing/Missing/' | sort -u
cat $OUT/vmem_counters_available.txt; echo
python3 tools/classes_pmc_summary.py $OUT $OUT/summary.csv && cat $OUT/summary.csv
