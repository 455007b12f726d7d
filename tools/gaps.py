"""Tuning aid: idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV.
python tools/gaps.py <kernel_trace.csv> [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
gap = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for a, b in zip(rows[:-1], rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    k = a["Kernel_Name"][:40] + " -> " + b["Kernel_Name"][:40]
    gap[k][0] += 1
    gap[k][1] += g
    busy += (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("kernels %d  span %.1f us  busy %.1f us  idle %.1f us (%.1f%%)" % (len(rows), span, busy, span - busy, 100 * (span - busy) / span))
for k, (n, t) in sorted(gap.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%6d x %8.2f us avg  %10.1f us total  %s" % (n, t / n, t, k))
