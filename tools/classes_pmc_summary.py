"""rocprofv3 --pmc counter_collection.csv files under <dir> -> one CSV row per kernel class: dispatches, the average of every
counter per dispatch and two ratios — the share of the kernel's duration the matrix pipes were busy (SQ_VALU_MFMA_BUSY_CYCLES
over 1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) and the issue-stall share of the waves' lifetime (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES).
SQ_INST_CYCLES_VMEM does not exist on gfx950 (rocprofv3: "Missing"): its _RD / _WR halves do, and are collected with
SQ_INSTS_VMEM_RD / _WR, SQ_ACTIVE_INST_VMEM and the LDS pair in a second pass."""
import collections
import csv
import glob
import re
import sys

CLASSES = [  # (class, regex over the kernel name as rocprofv3 prints it — demangled where its demangler copes —, source run)
    ("gemm_bf16_lnfold_qgelu_p256 (FC1)", r"gemm256p_kernel(ILi8E|<8,)", "bench"),
    ("gemm_bf16_lnfold_bias_p256 (QKV)", r"gemm256p_kernel(ILi7E|<7,)", "bench"),
    ("gemm_bf16_bias_resid16_p160_k3072 (FC2)", r"gemm160p_kernel(ILi9ELi0ELi48E|<9, 0, 48>)", "bench"),
    ("gemm_bf16_bias_resid16_p160_k768 (out-proj)", r"gemm160p_kernel(ILi9ELi0ELi12E|<9, 0, 12>)", "bench"),
    ("gemm_bf16_patch_p160", r"gemm160p_kernel(ILi4E|<4,)|gemm16_kernelIDF16bLi160ELi4E", "bench"),
    ("attention (B/32: 50 tokens, 4 heads per workgroup)", r"attention_heads_kernel", "bench"),
    ("score_gemm_f16 (step query, 256 x 100k)", r"gemm256s_kernelIDF16_|gemm(256|16)_kernelIDF16_", "bench"),
    ("gemm_bf16_lnfold_qgelu_p256 (L/14 FC1, bf16 leg, finished statistics)", r"gemm256p_kernel(ILi8ELi1E|<8, 1,)", "fp8"),
    ("gemm_bf16_lnfold_bias_p256 (L/14 QKV, bf16 leg, finished statistics)", r"gemm256p_kernel(ILi7ELi1E|<7, 1,)", "fp8"),
    ("gemm_bf16_bias_resid16 (L/14 out-proj, bf16 leg)", r"gemm16_kernelIDF16bLi1[0-9]+ELi9E", "fp8"),
    ("gemm_bf16_bias_resid16_p160_k4096 (L/14 FC2, bf16 leg)", r"gemm160p_kernel(ILi9ELi0ELi64E|<9, 0, 64>)", "fp8"),
    ("gemm8 (fp8 block-scaled GEMMs on the BM x 128 tile, round 4; L/14 bs 128)", r"gemm8_kernel", "fp8"),
    # round 5: the persistent 256 x 256 block-scaled fp8 kernel (gemm_fp8_p256.h), one class per epilogue
    ("gemm_fp8_bias_p256 (L/14 QKV, fp8)", r"gemm256p8_kernel(ILi0ELi0ELi0E|<0, 0, 0>)", "fp8"),
    ("gemm_fp8_qgelu_mx_p256 (L/14 FC1, fp8 -> MXFP8)", r"gemm256p8_kernel(ILi1ELi0ELi0E|<1, 0, 0>)", "fp8"),
    ("gemm_fp8_bias_resid16_p256 (L/14 out-proj + FC2, fp8)", r"gemm256p8_kernel(ILi3ELi0ELi0E|<3, 0, 0>)", "fp8"),
    ("attention_stream (L/14: 257 tokens, bf16 + MXFP8 output; round 5)", r"attention_(stream|long)_kernel", "fp8"),
    ("layernorm16 -> MXFP8 (L/14, 16 columns per lane)", r"layernorm16_mxfp8", "fp8"),
    # round 6: ViT-B/32 bs 256 under the fp8 setting (tools/b32_kernel_table.py): K = 768 = three K-tile pairs per tile (ODD form)
    ("gemm_fp8_bias_p256 (B/32 QKV, fp8, K = 768)", r"gemm256p8_kernel(ILi0ELi0ELi1E|<0, 0, 1>)", "b32fp8"),
    ("gemm_fp8_qgelu_mx_p256 (B/32 FC1, fp8 -> MXFP8, K = 768)", r"gemm256p8_kernel(ILi1ELi0ELi1E|<1, 0, 1>)", "b32fp8"),
    ("gemm_fp8_bias_resid16 (B/32 FC2 + out-proj, fp8 tile kernel)", r"gemm8_kernel", "b32fp8"),
    ("attention_mx (B/32: 50 tokens, MXFP8 output)", r"attention_heads_kernel", "b32fp8"),
    ("layernorm16 -> MXFP8 (B/32, d = 768)", r"layernorm16_mxfp8", "b32fp8"),
]
COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAVES",
            "GRBM_GUI_ACTIVE", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
            "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"]


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def main(root, out):
    runs = {"bench": load(root + "/bench"), "fp8": load(root + "/fp8"), "b32fp8": load(root + "/b32fp8")}
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["class", "run", "symbols", "dispatches"] + [c + "_per_dispatch" for c in COUNTERS] +
                   ["matrix_pipe_busy_share_of_kernel_time", "issue_stall_per_wave_cycle"])
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
            # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the SIMDs (= MFMAs x their pass cycles; exact on the GEMMs);
            # GRBM_GUI_ACTIVE: cycles, summed over the 8 XCDs -> kernel duration in cycles = / 8; 1024 SIMDs on the chip
            busy = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0)
            w.writerow([cls, run, len(names), n] + ["%.0f" % avg[c] for c in COUNTERS] +
                       ["%.3f" % busy, "%.3f" % (avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"])])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
