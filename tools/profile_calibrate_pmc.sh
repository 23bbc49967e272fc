cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gemm_pmc_calib; mkdir -p $OUT
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  n=$(echo $SET | tr ' ' '_')
  CALIB=1 ITERS=5 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/$n -o pmc -- python3 tools/gemm_p256_probe.py > $OUT/$n.log 2>&1
  echo "$SET rc=$?"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/gemm_pmc_calib/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "gemm256" not in kn and "copy" not in kn.lower() and "elementwise" not in kn.lower(): continue
        k = (kn[:50], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        print("%s %-28s %16.0f per dispatch (%d)" % (kn, cn, v / n, n))
PY
