"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into
profiles/r1/pmc_summary.json: per kernel, mean per-launch HBM bytes.

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B; on gfx950 this
rocprofv3 counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md,
section HBM), so the read side is doubled.  The write side is uncalibrated."""
import collections, csv, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_csv, write_csv = sys.argv[1], sys.argv[2]
out_path = os.path.join(root, "profiles", "r2", "pmc_summary.json") if len(sys.argv) < 4 else sys.argv[3]


def collect(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            kname = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            acc[kname.split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


f, w = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fk = sum(f[k]) / len(f[k]) if f.get(k) else 0.0
    wk = sum(w[k]) / len(w[k]) if w.get(k) else 0.0
    out[k] = {"launches_fetch_pass": len(f.get(k, [])), "launches_write_pass": len(w.get(k, [])),
              "fetch_size_kb_mean": fk, "write_size_kb_mean": wk,
              "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:10]:
    print("%-40s %10.1f MB/launch (fetch x2 %.1f, write %.1f)" % (k[:40], v["hbm_bytes_per_launch"] / 1e6,
          2 * v["fetch_size_kb_mean"] * 1024 / 1e6, v["write_size_kb_mean"] * 1024 / 1e6))
