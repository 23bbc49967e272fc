"""rocprofv3 --pmc counter_collection.csv files under <dir> -> one CSV row per kernel class: dispatches, the average of every
counter per dispatch and two ratios — matrix-pipe busy per wave lifetime (SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES, both
quad-cycle counts summed over the waves) and issue-stall share (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)."""
import collections
import csv
import glob
import re
import sys

CLASSES = [  # (class, regex over the mangled name, source run)
    ("gemm_bf16_lnfold_qgelu_p256 (FC1)", r"gemm256p_kernelILi8E", "bench"),
    ("gemm_bf16_lnfold_bias_p256 (QKV)", r"gemm256p_kernelILi7E", "bench"),
    ("gemm_bf16_bias_resid16_p160 (out-proj + FC2)", r"gemm160p_kernel", "bench"),
    ("gemm_bf16_bias_resid16 (L/14 out-proj, bf16 leg)", r"gemm16_kernelIDF16bLi1[0-9]+ELi9E", "fp8"),
    ("gemm_bf16_patch", r"gemm16_kernelIDF16bLi160ELi4E", "bench"),
    ("attention", r"attention_kernelILi2ELb0E", "bench"),
    ("score_gemm_f16 (step query, 256 x 100k)", r"gemm(256|16)_kernelIDF16_", "bench"),
    ("gemm_bf16_bias_qgelu_p256 (L/14 FC1, bf16 leg)", r"gemm256p_kernelILi2E", "fp8"),
    ("gemm_bf16_bias_p256 (L/14 QKV, bf16 leg)", r"gemm256p_kernelILi1E", "fp8"),
    ("gemm8 (fp8 block-scaled GEMMs, L/14 bs 128)", r"gemm8_kernel", "fp8"),
    ("attention (L/14: 257 tokens)", r"attention", "fp8"),
]
COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_LDS_BANK_CONFLICT",
            "SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"]


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def main(root, out):
    runs = {"bench": load(root + "/bench"), "fp8": load(root + "/fp8")}
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["class", "run", "symbols", "dispatches"] + [c + "_per_dispatch" for c in COUNTERS] +
                   ["mfma_busy_per_wave_cycle", "issue_stall_per_wave_cycle"])
        for cls, pat, run in CLASSES:
            rx = re.compile(pat)
            names = [k for k in runs[run] if rx.search(k)]
            if not names:
                continue
            avg = {}
            n = 0
            for c in COUNTERS:
                vals = [v for k in names for v in runs[run][k].get(c, [])]
                avg[c] = sum(vals) / len(vals) if vals else float("nan")
                n = max(n, len(vals))
            wc = avg["SQ_WAVE_CYCLES"]
            w.writerow([cls, run, len(names), n] + ["%.0f" % avg[c] for c in COUNTERS] +
                       ["%.3f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / wc), "%.3f" % (avg["SQ_WAIT_INST_ANY"] / wc)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
