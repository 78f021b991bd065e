#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 kernel trace (…_kernel_trace.csv): for every HSA queue the launches, the busy time (union of its
kernels' intervals), the idle time between its first and last kernel, and its ten heaviest kernels; then the union over all queues.
python tools/trace_streams.py <kernel_trace.csv> [skip_fraction]     skip_fraction (default 0.34) of the span from the start is left out
(the untimed first pass of tools/own_trained.py)."""
import csv
import sys
from collections import defaultdict


def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.34
    rows = list(csv.DictReader(open(path)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    cut = t0 + skip * (t1 - t0)
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
    span = (t1 - cut) / 1e6
    byq = defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append(r)
    print(f"{len(rows)} launches in the last {span:.1f} ms of the trace")
    alliv = []
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs]
        alliv += iv
        busy = union(list(iv)) / 1e6
        qs = (max(e for _, e in iv) - min(s for s, _ in iv)) / 1e6
        print(f"\nqueue {q}: {len(rs)} launches, busy {busy:.1f} ms = {busy / span:.3f} of the span, first-to-last {qs:.1f} ms")
        tot = defaultdict(lambda: [0, 0.0])
        for r in rs:
            n = r["Kernel_Name"].split("(")[0].split("<")[0][-48:]
            tot[n][0] += 1
            tot[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:10]:
            print(f"   {ms:9.2f} ms {c:7d}  {n}")
    print(f"\nall queues: union busy {union(alliv) / 1e6:.1f} ms = {union(alliv) / 1e6 / span:.3f} of the span")




def timeline(path):
    """Marker kernels of a trace, in time order: where each launch group's detector, NMS, filter and ReID rounds begin and end, and the busy
    stretches of the tracker kernels (launches less than 0.3 ms apart merged)."""
    marks = {"yolo_stem_fused": "YOLO begin", "select_sort_nms": "NMS", "det_filter_scatter": "filter", "reid_stem_pool2": "ReID begin",
             "avgpool8": "ReID end (avgpool)"}
    rows = list(csv.DictReader(open(path)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    ev, trk = [], []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "trk_epoch" in r["Kernel_Name"] or "gallery_commit" in r["Kernel_Name"]:
            trk.append((s, e))
        for k, lab in marks.items():
            if k in r["Kernel_Name"]:
                ev.append((s, e, lab, r["Queue_Id"], r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    trk.sort()
    cs = ce = None
    for s, e in trk:
        if cs is None:
            cs, ce = s, e
        elif s - ce < 300000:
            ce = max(ce, e)
        else:
            ev.append((cs, ce, "tracker busy", "-", "-"))
            cs, ce = s, e
    if cs is not None:
        ev.append((cs, ce, "tracker busy", "-", "-"))
    ev.sort()
    for s, e, lab, q, g in ev:
        print(f"  t = {(s - t0) / 1e6:9.2f} .. {(e - t0) / 1e6:9.2f} ms  queue {q}  {lab:20s} grid {g}")


if __name__ == "__main__":
    main()
    if len(sys.argv) > 3 and sys.argv[3] == "timeline":
        timeline(sys.argv[1])
