"""rocprofv3 --kernel-trace CSV -> per kernel: launches, mean duration, mean idle gap since the previous kernel ended."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 240       # the last launches only (steady state)
rows = rows[-tail:]
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
order = []
prev_end = None
for r in rows:
    nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    k = nm.split("(")[0][:44] + " g" + r.get("Grid_Size_X", r.get("Grid_Size", "?")) + " wg" + r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if k not in cnt: order.append(k)
    cnt[k] += 1; dur[k] += e - s
    if prev_end is not None: gap[k] += max(0, s - prev_end)
    prev_end = e if prev_end is None else max(prev_end, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} launches over {span / 1e3:.1f} us")
td = tg = 0
for k in order:
    print(f"{k:85s} x{cnt[k]:4d}  dur {dur[k] / cnt[k] / 1e3:7.2f} us  gap {gap[k] / cnt[k] / 1e3:6.2f} us")
    td += dur[k]; tg += gap[k]
print(f"busy {td / 1e3:.1f} us  idle {tg / 1e3:.1f} us")
