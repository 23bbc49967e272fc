#!/bin/bash
# HBM-side traffic of the bench step's kernels at HEAD's defaults: the FETCH_SIZE and WRITE_SIZE PMC passes, each in its OWN
# rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md, HBM section), aggregated by tools/traffic_from_pmc.py.
# Usage (through gpurun): bash tools/profile_pmc.sh r02
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
SHORT="--steps 5 --warmup 2 --retrieval-rows 0 --no-cpu-baseline --no-kernel-events --no-text"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 bench.py $SHORT > $OUT/fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- python3 bench.py $SHORT > $OUT/write.log 2>&1
echo "write rc=$?"
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/traffic_from_pmc.py "$F" "$W" $OUT/traffic.json && cat $OUT/traffic.json | head -60
# round 5: the same two passes over the second half of the metric (10M x 512 f16, Q = 1 scan and Q = 1024 score GEMM)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_retr -o pmc -- python3 tools/retrieval_profile.py > $OUT/fetch_retr.log 2>&1
echo "fetch retr rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_retr -o pmc -- python3 tools/retrieval_profile.py > $OUT/write_retr.log 2>&1
echo "write retr rc=$?"
F=$(find $OUT/fetch_retr -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write_retr -name "*counter_collection.csv" | head -1)
python3 tools/traffic_from_pmc.py "$F" "$W" $OUT/traffic_retrieval.json && cat $OUT/traffic_retrieval.json | head -40
