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
    # exposed time per family: the part of the step during which ONLY kernels of that family were running
    fam = lambda n: ("norm" if "norm_" in n else "wgrad" if "wgrad" in n else "winograd" if "wino" in n else "patch/narrow/flat" if "igemm" in n
                     else "pack" if "pack" in n else "aten/rocclr" if ("at::" in n or "rocclr" in n) else "other")
    ev = []
    for r in ks:
        ev.append((r[0], 1, fam(r[2])))
        ev.append((r[1], -1, fam(r[2])))
    ev.sort()
    active, excl, tprev = {}, {}, ev[0][0]
    for t, d, f in ev:
        live = [k for k, v in active.items() if v > 0]
        if len(live) == 1:
            excl[live[0]] = excl.get(live[0], 0) + (t - tprev)
        active[f] = active.get(f, 0) + d
        tprev = t
    print("   exposed (sole running family) ms: " + ", ".join("%s %.2f" % (k, v / 1e6) for k, v in sorted(excl.items(), key=lambda kv: -kv[1])))
    gaps.sort(reverse=True)
    print("   idle total %.2f ms; largest gaps:" % (sum(g[0] for g in gaps) / 1e6))
    for g in gaps[:6]:
        print("     %.3f ms  after %-50s before %s" % (g[0] / 1e6, g[1][:50], g[2][:50]))
