"""Pretty-print a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "grad_steps_per_sec", "grad_steps_per_sec_eager", "env_only_steps_per_sec", "ms_per_step")})
r = dict(d.get("roofline") or {})
rg = r.pop("replay_gather", None)
print("roofline:", r)
if rg:
    for k, v in rg.items():
        print("  gather", k, v)
for k in d.get("kernels", []):
    a = "" if k["achieved"] is None else f"{k['achieved']:9.2f} {k['unit']:8s} frac {k['frac']:.3f}"
    if "frac_issued" in k:
        a += f"  (issued {k['frac_issued']:.3f})"
    print(f"{k['kernel'][:60]:60s} {k['us']:8.2f} us  x{k['launches_per_step']}  {k['bound']:7s} {a}")
if d.get("cpu_baseline"):
    print("cpu:", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
