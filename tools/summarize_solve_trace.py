#!/usr/bin/env python3
"""Cut ONE solve out of a rocprofv3 kernel trace (between the last two qa_profile_marker_kernel dispatches, tools/mg_solve_profile.py),
join every dispatch with the algorithmic bytes the library recorded for that launch (i-th record of a kernel name <-> i-th dispatch of
that name) and print the table  kernel x tag x calls x time x bytes x fraction of the 8 TB/s HBM roofline, with the idle gaps between
dispatches as their own line, so that the lines sum to the measured window.

usage: summarize_solve_trace.py <kernel_trace.csv> <acct.json> <out.json> [solver_secs]"""
import collections
import csv
import json
import re
import sys

HBM = 8e12


def base(name):
    m = re.match(r"(?:void )?([\w:]+)", name)
    return m.group(1).split("::")[-1] if m else name


def main():
    trace, acct, out = sys.argv[1:4]
    solver_secs = float(sys.argv[4]) if len(sys.argv) > 4 else None
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "qa_profile_marker_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit("fewer than two marker dispatches in the trace")
    win = rows[marks[-2] + 1:marks[-1]]
    t_begin, t_end = int(rows[marks[-2]]["End_Timestamp"]), int(rows[marks[-1]]["Start_Timestamp"])
    recs = collections.defaultdict(list)
    for r in json.load(open(acct)):
        recs[r["kernel"]].append(r)
    seen = collections.Counter()
    table = collections.OrderedDict()
    busy, last_end, gaps = 0, t_begin, 0
    unmatched = collections.Counter()
    for r in win:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        b = base(r["Kernel_Name"])
        i = seen[b]
        seen[b] += 1
        rec = recs[b][i] if i < len(recs.get(b, [])) else None
        if rec is None and b in recs:
            unmatched[b] += 1
        key = (b, rec["tag"] if rec else "")
        e = table.setdefault(key, dict(kernel=b, tag=key[1], calls=0, ns=0, bytes=0.0, modelled=rec is not None))
        e["calls"] += 1
        e["ns"] += en - st
        if rec:
            e["bytes"] += rec["bytes"]
        busy += en - st
        if st > last_end:
            gaps += st - last_end
        last_end = max(last_end, en)
    window = t_end - t_begin
    lines = sorted(table.values(), key=lambda e: -e["ns"])
    for e in lines:
        e["ms"] = round(e["ns"] * 1e-6, 4)
        e["share_of_window"] = round(e["ns"] / window, 4)
        e["us_per_call"] = round(e["ns"] * 1e-3 / e["calls"], 2)
        if e["modelled"] and e["ns"]:
            e["GB"] = round(e["bytes"] * 1e-9, 4)
            e["TBps"] = round(e["bytes"] / (e["ns"] * 1e-9) * 1e-12, 3)
            e["frac_of_8TBps"] = round(e["bytes"] / (e["ns"] * 1e-9) / HBM, 4)
        del e["ns"], e["bytes"]
    summary = dict(window_ms=round(window * 1e-6, 3), kernels_busy_ms=round(busy * 1e-6, 3), idle_gaps_ms=round(gaps * 1e-6, 3), dispatches=len(win),
                   solver_secs_reported=solver_secs, accounting_records_without_dispatch={k: len(v) - seen[k] for k, v in recs.items() if len(v) != seen[k]},
                   note="window = end of the first marker dispatch to start of the second; idle gaps = time inside the window with no kernel running (host round trips of "
                        "the reductions, launch latency); bytes = ALGORITHMIC bytes recorded by the library per launch (qa_core.h acct), not counters")
    json.dump(dict(summary=summary, lines=lines), open(out, "w"), indent=1)
    print(json.dumps(summary))
    print("%-28s %-44s %6s %9s %7s %9s %8s %6s" % ("kernel", "tag", "calls", "ms", "share", "us/call", "GB", "frac"))
    for e in lines[:40]:
        print("%-28s %-44s %6d %9.3f %7.3f %9.2f %8s %6s" % (e["kernel"][:28], e["tag"][:44], e["calls"], e["ms"], e["share_of_window"], e["us_per_call"], e.get("GB", "-"), e.get("frac_of_8TBps", "-")))


if __name__ == "__main__":
    main()
