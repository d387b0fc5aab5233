#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [first_call] [ncalls] -- per-call timeline of the engine's launches from a
rocprofv3 --kernel-trace CSV: start / end of forward, mix, inverse (main stream) and the NCO chain (side stream)
relative to the call's forward start, gaps between launches, overlap of the chain with the launches."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = ("fwd" if "xlp_forward" in n else "mix" if "xlp_mix" in n else "inv" if "xlp_inverse" in n else
         "chain" if "xl_nco_chain" in n else "fir" if "xl_fir_kernel" in n else "nco" if "xl_nco_table" in n else None)
    if k:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Queue_Id"], int(r.get("VGPR_Count", 0) or 0)))
rows.sort()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ncalls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
starts = [i for i, r in enumerate(rows) if r[2] in ("fwd", "fir")]
if len(starts) <= first + ncalls:
    first = max(0, len(starts) - ncalls - 1)
t0 = rows[starts[first]][0]
print(f"{'kernel':6s} {'queue':>5s} {'start us':>10s} {'end us':>10s} {'dur us':>9s} {'gap to prev on queue':>22s}")
last_end = {}
for s, e, k, q, v in rows[starts[first]:starts[min(first + ncalls, len(starts) - 1)]]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print(f"{k:6s} {q:>5s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:22.1f}   vgpr {v}")
    last_end[q] = e
durs = {}
for s, e, k, q, v in rows[starts[first]:]:
    durs.setdefault(k, []).append((e - s) / 1e3)
print({k: round(sum(v) / len(v), 1) for k, v in durs.items()}, "mean us per launch;  calls:", len(starts))
per = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(starts[first:-1], starts[first + 1:])]
if per:
    print("call period us: mean %.1f min %.1f max %.1f" % (sum(per) / len(per), min(per), max(per)))
