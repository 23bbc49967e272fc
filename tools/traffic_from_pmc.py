"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each collected in its OWN run with --kernel-trace only)
into profiles/rNN_traffic.json: average HBM bytes per launch of the kernel classes bench.py's roofline reports.

    python tools/traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> profiles/r01_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB-like
units of 1024 B... the guide's gfx950 note: FETCH_SIZE under-reports by 2x (64 B requests counted as 32 B), so fetch
bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is taken as is. Cross-check: the flat scan of a 10M x 512 f16 index
reads 10.24 GB algorithmically and the corrected counter says 10.24 GB.
"""
import collections
import csv
import json
import re
import sys

# library kernel class -> regex over the (mangled or demangled) kernel name
CLASSES = {
    "gemm_bf16_bias_resid": r"gemm16_kernelIDF16bLi160ELi3E",
    "gemm_bf16_bias_resid16": r"gemm16_kernelIDF16bLi160ELi9E",
    # round 4: one symbol per K (gemm160p_kernel<9, 0, K/64>): out-projection and FC2 of the bs-256 step apart
    "gemm_bf16_bias_resid16_p160_k768": r"gemm160p_kernel(ILi9ELi0ELi12E|<9, 0, 12>)",
    "gemm_bf16_bias_resid16_p160_k3072": r"gemm160p_kernel(ILi9ELi0ELi48E|<9, 0, 48>)",
    "gemm_bf16_patch_p160": r"gemm160p_kernel(ILi4E|<4,)",
    "gemm_bf16_bias_resid_pruned": r"gemm16_kernelIDF16bLi128ELi3E",
    "gemm_bf16_bias_qgelu": r"gemm16_kernelIDF16bLi192ELi2E",
    "gemm_bf16_lnfold_qgelu": r"gemm16_kernelIDF16bLi192ELi8E",
    "gemm_bf16_lnfold_bias": r"gemm16_kernelIDF16bLi192ELi7E|gemm256_kernelIDF16bLi7E",
    "gemm_bf16_bias": r"gemm16_kernelIDF16bLi192ELi1E",
    # round 3: the persistent 256 x 256 kernel (gemm_bf16_p256.h) takes the folded QKV / FC1 GEMMs of the bs-256 step
    "gemm_bf16_lnfold_qgelu_p256": r"gemm256p_kernel(ILi8E|<8,)",
    "gemm_bf16_lnfold_bias_p256": r"gemm256p_kernel(ILi7E|<7,)",
    "gemm_bf16_bias_qgelu_p256": r"gemm256p_kernel(ILi2E|<2,)",
    "gemm_bf16_bias_p256": r"gemm256p_kernel(ILi1E|<1,)",
    "score_gemm_f16_strip": r"gemm256s_kernelIDF16_",
    "gemm_bf16_patch": r"gemm16_kernelIDF16bLi160ELi4E",
    "score_gemm_f16": r"gemm256_kernelIDF16_Li5E",
    "scan_topk_f16": r"scan_topk_kernelIDF16_",
    "layernorm": r"layernorm_kernel(<true>|ILb1E)",
    "attention": r"attention(_heads|_long)?_kernel(<|ILi)",
    "im2col": r"im2col_kernel(<|ILb)",
}
ALGORITHMIC = {  # bytes per launch the kernel must move, averaged over the shapes in the class
    # residual GEMM in the folded-LayerNorm mode (the default from 6000 rows): 130.7 MB of GEMM traffic + the bf16 copy of
    # the new residual rows (19.7 MB) + their partial LayerNorm statistics (1.2 MB) that replace the LayerNorm pass
    # bf16 residual stream: out-proj 19.7 (A) + 1.2 (W) + 2 x 19.7 (stream) + 1.2 (stats) = 61.5 MB, FC2 78.6 + 4.7 + 39.3 + 1.2
    # = 123.8 MB -> class average 92.7 MB
    "gemm_bf16_bias_resid16": 92.7e6, "gemm_bf16_bias_resid16_p160_k768": 61.5e6, "gemm_bf16_bias_resid16_p160_k3072": 123.8e6,
    "gemm_bf16_bias_resid": 151.6e6, "gemm_bf16_bias_qgelu": 103.0e6, "gemm_bf16_bias": 82.2e6,
    "gemm_bf16_lnfold_qgelu": 104.3e6, "gemm_bf16_lnfold_bias": 83.5e6,
    # FC1 12800 x 3072 x 768: 19.7 (A) + 4.7 (W) + 78.6 (out) + 1.2 (row statistics) MB; QKV x 2304: 19.7 + 3.5 + 59.0 + 1.2
    "gemm_bf16_lnfold_qgelu_p256": 104.3e6, "gemm_bf16_lnfold_bias_p256": 83.5e6,
    "scan_topk_f16": 10.24e9,
}


def retrieval_passes():
    """Query batches per Q leg of tools/retrieval_profile.py (the run the retrieval PMC passes are collected over)."""
    import os
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "retrieval_profile.py")).read()
    m = re.search(r"^PASSES\s*=\s*(\d+)", text, flags=re.M)
    return int(m.group(1)) if m else None


def per_kernel(path):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out):
    fetch, write = per_kernel(fetch_csv), per_kernel(write_csv)
    res = {"_note": "HBM bytes per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, "
                    "--kernel-trace only), FETCH_SIZE doubled as MI355X_MICROARCH.md 'HBM' prescribes for gfx950; command: "
                    "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 5 --warmup 2 "
                    "--retrieval-rows 0 --no-cpu-baseline --no-kernel-events --no-text (tools/profile_pmc.sh); Infinity-Cache "
                    "hits are counted: fabric traffic, an upper bound on HBM traffic; aggregated by tools/traffic_from_pmc.py"}
    for cls, pat in CLASSES.items():
        rx = re.compile(pat)
        fv = [v for k, vs in fetch.items() if rx.search(k) for v in vs]
        wv = [v for k, vs in write.items() if rx.search(k) for v in vs]
        if not fv or not wv:
            continue
        fb = 2.0 * 1024.0 * sum(fv) / len(fv)
        wb = 1024.0 * sum(wv) / len(wv)
        ent = {"symbol_regex": pat, "launches_sampled": len(fv), "fetch_bytes_corrected": round(fb),
               "write_bytes": round(wb), "traffic_bytes": round(fb + wb),
               # all sampled launches together (classes whose launches differ in size: the score GEMM's sample and filtered pass)
               "traffic_bytes_all_launches": round(2.0 * 1024.0 * sum(fv) + 1024.0 * sum(wv))}
        if cls in ("score_gemm_f16_strip", "scan_topk_f16") and retrieval_passes():
            # (only meaningful for passes collected over tools/retrieval_profile.py: bench.py divides the class total by it)
            ent["batches_if_over_retrieval_profile"] = retrieval_passes()
        if cls in ALGORITHMIC:
            ent["algorithmic_bytes_avg"] = ALGORITHMIC[cls]
            ent["ratio_to_algorithmic"] = round((fb + wb) / ALGORITHMIC[cls], 3)
        res[cls] = ent
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
