#!/bin/bash
# A/B on the GPU box: the persistent fp8 GEMM with a phase's two LDS-DMA pieces issued from the middle of its MFMA part
# (-DQ256_STAGE_MID) against the default (pieces in the read part). usage (gpurun): bash tools/q256_stagemid_ab.sh
set -u
cd "$GRAFT_REPO_ROOT"
trap 'make -C multimodal-image-similarity-search_amd/csrc clean > /dev/null; make -C multimodal-image-similarity-search_amd/csrc -j16 > gpurun_out/ab_restore.log 2>&1 || tail -5 gpurun_out/ab_restore.log' EXIT
export MMISS_ALLOW_AB_BUILD=1
run() { timeout -k 10 200 python tools/gemm8p_probe.py 2>&1 | grep -v amdgpu | sed 's/bm128.*bm256:/bm256:/'; }
echo "== default"
run
echo "== -DQ256_STAGE_MID"
make -C multimodal-image-similarity-search_amd/csrc clean > /dev/null
make -C multimodal-image-similarity-search_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -DQ256_STAGE_MID" > gpurun_out/ab_build.log 2>&1 || tail -5 gpurun_out/ab_build.log
timeout -k 10 400 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q -k "bit_for_bit or identical_bytes or folded_in or leaves_the_rows" 2>&1 | tail -2
run
timeout -k 10 200 python tools/l14_fp8_ab.py 2>&1 | grep -E "^base"
