"""HBM bytes per launch from the two rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE; separate passes as
MI355X_MICROARCH.md prescribes): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads (the factor 2; uncalibrated for narrow loads, so the read side is an upper bound).  Writes gpurun_out/<tag>_traffic.json
(keyed by bench.py's kernel labels) and gpurun_out/<tag>_pmc_fetch_write_raw.json (per kernel name); copy both into profiles/."""
import collections, csv, glob, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def mean_counter(dirpat, cname):
    import os
    f = max(glob.glob(dirpat + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)      # (the newest pass: gpurun_out/ keeps earlier ones)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[name + "@" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fe, wr = mean_counter(f"gpurun_out/{tag}_pmc_fetch", "FETCH_SIZE"), mean_counter(f"gpurun_out/{tag}_pmc_write", "WRITE_SIZE")
raw = {k: {"FETCH_SIZE_KB": fe[k], "WRITE_SIZE_KB": wr.get(k, 0.0), "hbm_bytes": int((2 * fe[k] + wr.get(k, 0.0)) * 1024)} for k in sorted(fe)}
json.dump(raw, open(f"gpurun_out/{tag}_pmc_fetch_write_raw.json", "w"), indent=1)


def pick(prefix, grid_pred=lambda g: True):
    for k, v in raw.items():
        name, grid = k.split("@")
        if name.startswith(prefix) and grid_pred(int(grid) if grid.isdigit() else 0):
            return v["hbm_bytes"]
    return None


# bench.py's labels -> (kernel name prefix, grid filter where one kernel appears at two sizes)
small = lambda g: g < 200000
t = {"_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, two separate rocprofv3 --pmc passes (tools/profile_round.sh); "
              "FETCH_SIZE halves wide coalesced reads on gfx950 (MI355X_MICROARCH.md), hence the factor 2; upper bound for narrow loads.",
     "conv23_sp_kernel<C1>[conv1 + pool + conv2 + conv3, 5 states per workgroup][act n=1024]": pick("conv23_sp_kernel<3, 5, true>"),
     "conv1_sp_kernel<nib>[act n=1024]": pick("conv1_sp_kernel<true>"), "conv23_sp_kernel[act n=1024]": pick("conv23_sp_kernel<3, 5, false>"),
     "fc1_sp_kernel[act n=1024]": pick("fc1_sp_kernel<3>"), "head_kernel[act n=1024]": pick("head_kernel"),
     "env_kernel<true>[n=1024]": pick("env_kernel<true>"),
     "conv1_pool_kernel[train 2B=64, gathered minibatch]": pick("conv1_pool_kernel<false>"),
     "conv23_t_kernel[train 2B=64, gathered minibatch]": pick("conv23_t_kernel<3, false"),
     "conv23_t_kernel<ring>[train 2B=64: conv1 + pool + conv2 + conv3 from the frame ring]": pick("conv23_t_kernel<3, true"),
     "fc1_fk_kernel[train 2B=64]": pick("fc1_fk_kernel"),
     "fc1_bwd2_kernel": pick("fc1_bwd2_kernel"),
     "conv_bx_kernel[conv3^T / conv2^T chain + conv3 dW + Adam of W_fc1 (HBM part priced)]": pick("conv_bx_kernel<3>"),
     "conv_dw21_kernel[gathered minibatch]": pick("conv_dw21_kernel<2, false>"), "conv_dw21_kernel<ring>": pick("conv_dw21_kernel<2, true>"),
     "conv_bw_kernel[gathered minibatch: per-sample conv3^T / conv2^T chain + conv3 / conv2 / conv1 dW + Adam of W_fc1 (HBM part priced)]": pick("conv_bw_kernel<3, false>"),
     "conv_bw_kernel<ring>[per-sample conv3^T / conv2^T chain + conv3 / conv2 / conv1 dW; Adam of W_fc1 rides (22.9 MB)]": pick("conv_bw_kernel<3, true>"),
     "adam_fused_kernel[all but W_fc1; emits the conv planes]": pick("adam_fused_kernel"),
     "gather_kernel<false>[B=32]": pick("gather_kernel<false>", lambda g: g < 100000),
     "gather_kernel<false>[B=256]": pick("gather_kernel<false>", lambda g: 100000 < g < 1000000),
     "gather_kernel<false>[B=4096]": pick("gather_kernel<false>", lambda g: g > 1000000)}
json.dump(t, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(t, indent=1))
