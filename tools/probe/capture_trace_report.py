"""Summarise a trace written by tools/probe/hip_capture_trace.cpp (LD_PRELOAD): which streams took part in the capture, every
wait with the stream its event was recorded on, and the anomalies: a stream waiting on its own event (on the origin stream / on
a forked stream), a wait on an event never recorded inside the capture, a stream that was forked and not joined back."""
import sys
from collections import Counter, defaultdict


def main(path):
    names, rec_on, rec_seq = {}, {}, {}
    work = Counter()
    capturing, origin = set(), None
    waits, anomalies = Counter(), []
    last_work_seq, joined_at = defaultdict(int), {}
    seq = 0
    tids = Counter()
    dead_recorded = []

    def nm(s):
        if s not in names:
            names[s] = "S%d" % len(names)
        return names[s]
    for line in open(path):
        p = line.split()
        if not p:
            continue
        seq += 1
        k = p[0]
        if k == "B":
            origin = p[1]
            capturing.add(origin)
            nm(origin)
        elif k in ("K", "M"):
            s = p[1]
            work[nm(s)] += 1
            last_work_seq[s] = seq
            tids[(nm(s), p[3])] += 1
            if s not in capturing:
                anomalies.append("seq %d: work on %s which is not part of the capture" % (seq, nm(s)))
        elif k == "R":
            ev, s = p[1], p[2]
            rec_on[ev], rec_seq[ev] = s, seq
            nm(s)
        elif k == "W":
            s, ev = p[1], p[2]
            src = rec_on.get(ev)
            if src is None:
                anomalies.append("seq %d: %s waits on an event never recorded inside the capture" % (seq, nm(s)))
                continue
            waits[(nm(s), nm(src))] += 1
            if src == s:
                anomalies.append("seq %d: SELF-WAIT on %s (%s stream), event recorded at seq %d, work on the stream since: %s, thread %s"
                                 % (seq, nm(s), "origin" if s == origin else "forked", rec_seq[ev], last_work_seq[s] > rec_seq[ev], p[4]))
            if src in capturing:
                capturing.add(s)
            if s == origin:
                joined_at[src] = seq
        elif k == "D":
            if p[1] in rec_on:
                dead_recorded.append(p[1])
        elif k == "E":
            for s in capturing:
                if s != origin and last_work_seq[s] > joined_at.get(s, 0):
                    anomalies.append("at EndCapture: %s has work after its last join into the origin stream (seq %d > %d)" % (nm(s), last_work_seq[s], joined_at.get(s, 0)))
    print("streams:", {v: k for k, v in names.items()})
    print("origin:", nm(origin) if origin else None, " participating:", sorted(nm(s) for s in capturing))
    print("work items per stream:", dict(work))
    print("issuing threads per stream:", dict(tids))
    print("waits (waiter <- recorded on):", dict(waits))
    print("events destroyed during the capture after having been recorded in it: %d" % len(dead_recorded))
    print("%d anomalies" % len(anomalies))
    for a in anomalies[:60]:
        print("  ", a)


if __name__ == "__main__":
    main(sys.argv[1])
