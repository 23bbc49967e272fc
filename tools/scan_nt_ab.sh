set -u
cd "$GRAFT_REPO_ROOT"
# The variant is built IN PLACE (the tools load multimodal-image-similarity-search_amd/libmmiss.so); whatever happens, the product build is
# restored on exit, and while the variant is in place mmiss_amd refuses to load it without MMISS_ALLOW_AB_BUILD=1.
trap 'make -C multimodal-image-similarity-search_amd/csrc clean > /dev/null; make -C multimodal-image-similarity-search_amd/csrc -j16 > gpurun_out/ab_restore.log 2>&1 || tail -5 gpurun_out/ab_restore.log' EXIT
export MMISS_ALLOW_AB_BUILD=1
run() { for i in 1 2; do python tools/single_request.py 400 2>&1 | grep -v amdgpu; done; }
echo "== plain row loads (default)"
run
echo "== non-temporal row loads in the scan (-DMMISS_SCAN_NT)"
rm -f multimodal-image-similarity-search_amd/csrc/api_index.o
make -C multimodal-image-similarity-search_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -DMMISS_SCAN_NT" > gpurun_out/ab_build.log 2>&1 || tail -5 gpurun_out/ab_build.log
run
