"""Pretty-print a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "grad_steps_per_sec", "grad_steps_per_sec_eager", "env_only_steps_per_sec", "ms_per_step")})
print("roofline:", d.get("roofline"))
for k in d.get("kernels", []):
    print(f"{k['kernel']:45s} {k['us']:8.2f} us  {k['achieved']:9.2f} {k['unit']:8s} frac {k['frac']:.3f}")
if d.get("cpu_baseline"):
    print("cpu:", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
