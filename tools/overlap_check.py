"""From a rocprofv3 --kernel-trace CSV of the pipelined bench: do the query-stage kernels run beside encoder kernels?
Per query kernel class: mean duration and the mean fraction of it during which some encoder kernel was also executing."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.3):int(len(rows) * 0.6)]
QUERY = ("prep_queries", "gemm256_kernelIDF16_", "gemm16_kernelIDF16_", "gemm256s", "select_topk", "merge_lists", "rerank", "scan_topk")
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
enc = [(s, e) for s, e, n, q in iv if not any(k in n for k in QUERY)]
st = collections.defaultdict(lambda: [0, 0.0, 0.0, set()])
for s, e, n, q in iv:
    if not any(k in n for k in QUERY):
        continue
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in enc if s2 < e and e2 > s)
    x = st[n[:48]]
    x[0] += 1; x[1] += (e - s) / 1e3; x[2] += ov / max(1, e - s); x[3].add(q)
for n, (c, d, o, q) in st.items():
    print("%-48s n=%3d  avg %7.1f us  overlapped with encoder kernels %.0f %%  queues %s" % (n, c, d / c, 100 * o / c, sorted(q)))
span = iv[-1][1] - iv[0][0]
busy_union = 0
cur_s, cur_e = None, None
for s, e, n, q in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy_union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy_union += cur_e - cur_s
print("window span %.0f us, union of kernel intervals %.0f us, sum of durations %.0f us" % (span / 1e3, busy_union / 1e3, sum(e - s for s, e, n, q in iv) / 1e3))
