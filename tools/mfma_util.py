"""MFMA utilisation per kernel from a rocprofv3 --pmc pass that holds SQ_VALU_MFMA_BUSY_CYCLES and SQ_INSTS_MFMA (tools/pmc_act.sh,
tools/pmc_train.sh):  utilisation = busy cycles (summed over the chip's 1024 SIMDs) / (1024 x kernel duration x 2.4 GHz nominal clock).
    python tools/mfma_util.py <counter_collection.csv> [...]"""
import collections, csv, sys

SIMDS, CLK = 256 * 4, 2.4e9
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9)
    for n, v in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or n.startswith("__amd") or n.startswith("at::"):
            continue
        busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
        insts = sum(v.get("SQ_INSTS_MFMA", [0])) / max(1, len(v.get("SQ_INSTS_MFMA", [0])))
        d = sum(dur[n]) / len(dur[n])
        if busy == 0:
            continue
        print(f"{n[:44]:44s} {d * 1e6:7.1f} us (under the profiler)  MFMA insts {insts:9.3g}  busy cycles {busy:9.3g} "
              f"({busy / max(insts, 1):4.1f} per inst)  utilisation {busy / (SIMDS * d * CLK):6.1%} of the matrix pipes at 2.4 GHz")
