#!/bin/bash
# One-call profile set of a round (run on the GPU box through gpurun): rocprofv3 kernel stats of the bench step and
# the FETCH_SIZE / WRITE_SIZE PMC passes (each in its OWN run, --kernel-trace only) for the tile orders under test.
# Usage: bash tools/profile_round.sh <round tag, e.g. r02>
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
SHORT="--steps 5 --warmup 2 --retrieval-rows 0 --no-cpu-baseline --no-kernel-events --no-text"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 20 --warmup 5 --retrieval-rows 0 --no-cpu-baseline --no-kernel-events --no-text > $OUT/stats.log 2>&1
echo "stats rc=$?"
for G in 0 5 10; do
  MMISS_OPTIONS=gemm_group_m=$G rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_g$G -o pmc -- python3 bench.py $SHORT > $OUT/fetch_g$G.log 2>&1
  echo "fetch g=$G rc=$?"
  MMISS_OPTIONS=gemm_group_m=$G rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_g$G -o pmc -- python3 bench.py $SHORT > $OUT/write_g$G.log 2>&1
  echo "write g=$G rc=$?"
done
find $OUT -name "*.csv" | head -40
du -sh $OUT
