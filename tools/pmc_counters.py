#!/usr/bin/env python3
"""Per-kernel averages of the counters of ONE rocprofv3 --pmc pass (csv output) for the kernels whose name contains a filter:

    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS \\
        --kernel-trace --output-format csv -d /tmp/pmc_sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline --eager
    python tools/pmc_counters.py /tmp/pmc_sq/sq_counter_collection.csv attn_ profiles/rNN_attn_pmc.json

Per kernel: dispatches, every counter per dispatch, and the derived ratios the attention analysis quotes (VALU instructions per MFMA
instruction, share of active cycles that issue VALU). SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import json
import sys


def main():
    path, filt, out = sys.argv[1:4]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if filt in k:
            a = agg[k.split("(")[0]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    res = {}
    for k, cs in agg.items():
        d = {"dispatches": max(v[0] for v in cs.values())}
        for c, (n, s) in cs.items():
            d[c] = s / n
        if d.get("SQ_INSTS_MFMA"):
            d["valu_per_mfma"] = d.get("SQ_INSTS_VALU", 0.0) / d["SQ_INSTS_MFMA"]
        if d.get("SQ_ACTIVE_INST_ANY"):
            d["valu_share_of_active_inst_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_ACTIVE_INST_ANY"]
        if d.get("SQ_WAVE_CYCLES"):
            d["active_share_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if d.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0.0) / d["SQ_WAVES"]
            d["mfma_insts_per_wave"] = d.get("SQ_INSTS_MFMA", 0.0) / d["SQ_WAVES"]
        res[k] = d
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
