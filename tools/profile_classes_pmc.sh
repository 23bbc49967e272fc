#!/bin/bash
# Matrix-pipe / wait / memory-issue counters per kernel CLASS of the round's hot paths, one pass each over
#   (a) the bench step at its defaults (ViT-B/32 bs 256, bf16: the four GEMM classes, attention, the score GEMM of the step's query) and
#   (b) one ViT-L/14 bs-128 encode in fp8 (gemm8_kernel) — tools/l14_fp8_bench.py,
# each rocprofv3 run with --kernel-trace + --pmc only (gpurun refuses --pmc next to the API trace domains).
# Usage (through gpurun): bash tools/profile_classes_pmc.sh r03   ->  gpurun_out/classes_pmc_r03/summary.csv
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/classes_pmc_$TAG
mkdir -p $OUT
SET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
SHORT="--steps 5 --warmup 2 --retrieval-rows 0 --no-cpu-baseline --no-kernel-events --no-text"
rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/bench -o pmc -- python3 bench.py $SHORT > $OUT/bench.log 2>&1
echo "bench pass rc=$?"
rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/fp8 -o pmc -- python3 tools/l14_fp8_bench.py > $OUT/fp8.log 2>&1
echo "fp8 pass rc=$?"
python3 tools/classes_pmc_summary.py $OUT $OUT/summary.csv && cat $OUT/summary.csv
