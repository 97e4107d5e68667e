"""profiles/traffic.json (HBM bytes per launch, keyed by bench.py's kernel labels) from the two rocprofv3 --pmc passes of
tools/pmc_passes.sh: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE reads half on gfx950)."""
import collections, csv, glob, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01i"


def mean_counter(dirpat, cname):
    f = sorted(glob.glob(dirpat + "/**/*counter_collection.csv", recursive=True))[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fe, wr = mean_counter(f"gpurun_out/{tag}_pmc_fetch", "FETCH_SIZE"), mean_counter(f"gpurun_out/{tag}_pmc_write", "WRITE_SIZE")
json.dump({k: {"FETCH_SIZE_KB": fe[k], "WRITE_SIZE_KB": wr.get(k, 0.0)} for k in fe}, open("profiles/r01_i_pmc_fetch_write_raw.json", "w"), indent=1)
t = json.load(open("profiles/traffic.json"))
labels = {"conv1_sp_kernel<nib>[act n=1024]": "conv1_sp_kernel<true>", "conv23_sp_kernel[act n=1024]": "conv23_sp_kernel<3>",
          "fc1_sp_kernel[act n=1024]": "fc1_sp_kernel<3>", "head_kernel[act n=1024]": "head_kernel", "env_kernel<true>[n=1024]": "env_kernel<true>",
          "adam_kernel[all but W_fc1]": "adam_kernel", "conv1_pool_kernel[train 2B=64]": "conv1_pool_kernel<false>", "conv2_kernel[train 2B=64]": "conv2_kernel",
          "conv3_kernel[train 2B=64]": "conv3_kernel", "fc1_kernel[train 2B=64]": "fc1_kernel", "loss_head_kernel": "loss_head_kernel",
          "fc1_bwd_kernel": "fc1_bwd_kernel", "conv3_bwd_kernel[+ Adam of W_fc1]": "conv3_bwd_kernel", "conv2_bwd_kernel": "conv2_bwd_kernel",
          "conv1_dw_kernel": "conv1_dw_kernel"}
for stale in ("adam_kernel", "conv3_bwd_kernel"):
    t.pop(stale, None)
for label, k in labels.items():
    t[label] = int((2 * fe[k] + wr[k]) * 1024)
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
print({k: t[k] for k in labels})
