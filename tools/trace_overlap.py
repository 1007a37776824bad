"""From a rocprofv3 --kernel-trace CSV of bench.py: per train step, wall time, union of kernel intervals (GPU busy), sum of kernel
durations, and the largest idle gaps with the kernels around them.  usage: python tools/trace_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")))
rows.sort()
# steps are delimited by the AdamW kernel launches (two per step: G then D)
adam = [i for i, r in enumerate(rows) if "adamw" in r[2]]
bounds = [rows[adam[i]][1] for i in range(1, len(adam), 2)]          # end of every second AdamW = end of a step
for s in range(len(bounds) - 1):
    t0, t1 = bounds[s], bounds[s + 1]
    ks = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    if not ks:
        continue
    busy, cur_s, cur_e, gaps = 0, ks[0][0], ks[0][1], []
    last = ks[0]
    for r in ks[1:]:
        if r[0] > cur_e:
            busy += cur_e - cur_s
            gaps.append((r[0] - cur_e, last[2], r[2]))
            cur_s, cur_e = r[0], r[1]
        else:
            cur_e = max(cur_e, r[1])
        if r[1] >= cur_e:
            last = r
    busy += cur_e - cur_s
    tot = sum(r[1] - r[0] for r in ks)
    queues = sorted({r[3] for r in ks})
    print("step %d: wall %.2f ms, GPU busy (union) %.2f ms, sum of kernels %.2f ms (x%.2f), %d kernels, queues %s" % (
        s, (t1 - t0) / 1e6, busy / 1e6, tot / 1e6, tot / max(busy, 1), len(ks), queues))
    gaps.sort(reverse=True)
    print("   idle total %.2f ms; largest gaps:" % (sum(g[0] for g in gaps) / 1e6))
    for g in gaps[:6]:
        print("     %.3f ms  after %-50s before %s" % (g[0] / 1e6, g[1][:50], g[2][:50]))
