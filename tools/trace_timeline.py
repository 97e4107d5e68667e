"""rocprofv3 --kernel-trace CSV -> the last N launches as a timeline (start / end in us relative to the first, queue id): which kernels overlap."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"q{r.get('Queue_Id', '?'):>3} {s:9.2f} {e:9.2f}  {e - s:7.2f}  {nm}")
