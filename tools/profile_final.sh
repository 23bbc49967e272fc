#!/bin/bash
# Final artefact set of a round, one call on one box: GPU test suite, smoke(), the default bench line, and the rocprofv3
# kernel-trace summary of the same step. Usage (through gpurun): bash tools/profile_final.sh r02
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 20 --warmup 5 --retrieval-rows 0 --no-cpu-baseline --no-kernel-events --no-text > $OUT/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_retr -o retr -- python3 tools/retrieval_profile.py > $OUT/stats_retr.log 2>&1; echo "retr stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b32fp8 -o b32fp8 -- python3 tools/b32_kernel_table.py > $OUT/stats_b32fp8.log 2>&1; echo "b32 fp8 stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l14 -o l14 -- python3 tools/l14_fp8_bench.py > $OUT/stats_l14.log 2>&1; echo "l14 stats rc=$?"
ls $OUT $OUT/stats | head -30
