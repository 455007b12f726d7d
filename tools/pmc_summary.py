"""Mean of each hardware counter per kernel from rocprofv3 --pmc CSV output.
python tools/pmc_summary.py out.json counter_collection.csv [more.csv ...]"""
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
out = {k: {c: [v[0] / v[1], v[1]] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    print(k[:70], {c: round(v[0], 1) for c, v in d.items()})
