"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    n = n.split('(')[0][:28]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in agg.items():
    if n.startswith('__amd') or n.startswith('at::'):
        continue
    d = {c: sum(x) / len(x) for c, x in v.items()}
    print(f"{n:28s} {sum(dur[n])/len(dur[n]):8.1f} us  " + "  ".join(f"{c}={x:.3g}" for c, x in sorted(d.items())))
