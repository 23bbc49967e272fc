set -u
cd "$GRAFT_REPO_ROOT"
# Wave priority (s_setprio) inside / outside the MFMA part of a phase: -DP256_PRIO = 4 * inside + outside.
# Variants are built IN PLACE; the product build is restored on exit.
trap 'make -C multimodal-image-similarity-search_amd/csrc clean > /dev/null; make -C multimodal-image-similarity-search_amd/csrc -j16 > gpurun_out/ab_restore.log 2>&1 || tail -5 gpurun_out/ab_restore.log' EXIT
export MMISS_ALLOW_AB_BUILD=1
run() { for i in 1 2; do python tools/gemm_p256_probe.py; EPI=7 N=2304 python tools/gemm_p256_probe.py; M=4096 N=4096 K=4096 EPI=1 ITERS=20 python tools/gemm_p256_probe.py; done 2>&1 | grep -v amdgpu; }
echo "== priority 1 in the MFMA part, 0 outside (default)"
run
for V in "-DP256_PRIO=0" "-DP256_PRIO=1" "-DP256_PRIO=12" "-DP256_PRIO=13"; do
  echo "== $V"
  make -C multimodal-image-similarity-search_amd/csrc clean > /dev/null
  make -C multimodal-image-similarity-search_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable $V" > gpurun_out/ab_build.log 2>&1 || tail -5 gpurun_out/ab_build.log
  timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "p256" 2>&1 | tail -1
  run
done
