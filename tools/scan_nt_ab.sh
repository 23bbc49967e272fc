set -u
cd "$GRAFT_REPO_ROOT"
run() { for i in 1 2; do python tools/single_request.py 400 2>&1 | grep -v amdgpu; done; }
echo "== plain row loads (default)"
run
echo "== non-temporal row loads in the scan (-DMMISS_SCAN_NT)"
rm -f multimodal-image-similarity-search_amd/csrc/api_index.o
make -C multimodal-image-similarity-search_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -DMMISS_SCAN_NT" > gpurun_out/ab_build.log 2>&1 || tail -5 gpurun_out/ab_build.log
run
