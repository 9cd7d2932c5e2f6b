"""Reduce one rocprofv3 --pmc pass (SQ counters) to per-kernel means: profiles/r2/mfma_counters.json.
Per kernel: launches seen, mean of every counter per launch, and the derived ratios the DESIGN quotes
(MFMA-busy share of the busy CU cycles, MFMA f64 ops per launch)."""
import collections, csv, json, sys

src, out_path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    d = {"launches": max(len(v) for v in cs.values())}
    for c, v in cs.items():
        d[c] = sum(v) / len(v)
    if d.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_over_busy_cu_cycles"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / d["SQ_BUSY_CU_CYCLES"]
    if d.get("SQ_WAVE_CYCLES"):
        d["valu_active_over_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"]
    out[k] = d
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0.0) * kv[1]["launches"])[:12]:
    print("%-44s launches %5d  mfma_busy/busy_cu %.3f  valu_active/wave_cycles %.3f  mfma insts %.0f" % (
        k[:44], d["launches"], d.get("mfma_busy_over_busy_cu_cycles", float("nan")),
        d.get("valu_active_over_wave_cycles", float("nan")), d.get("SQ_INSTS_MFMA", d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0))))
