# kernel timeline of steady-state EM iterations of the bench workload around the two seams of an iteration:
# E-step's last launch -> H-step's first round, and H-step's last round -> the next E-step's first launch
# (rocprofv3 --kernel-trace; the tracer stretches concurrent launches, the serial seams are what to read here)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/iter_tl; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --steps 8 --warmup 6 --no-cpu-baseline > $O/log.txt 2>&1
T=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("esplit_ya", "esplit_pass", "esplit_lane", "esplit_latent", "esplit_mix", "esplit_cols", "hstep_round_lr", "hstep_round_mfma",
              "hstep_lr_tables", "hstep_moment_kernel", "hstep_moment_reduce", "hstep_w_latent_major", "ichol_exact_wave", "mstep_accum",
              "mstep_sum_solve", "latent_moments", "sum_partials", "latent_map", "noise_stats", "copyBuffer", "fillBuffer", "transpose"):
        if k in n:
            return k
    return n[:40]
ya = [i for i, r in enumerate(rows) if "esplit_ya" in r["Kernel_Name"]]
# the seam around the third-last E-step start
for which in (-3, -2):
    i0 = ya[which]
    t0 = int(rows[i0]["Start_Timestamp"])
    print("---- seam: H-step's last rounds -> E-step (t = 0 at esplit_ya start)")
    for r in rows[max(0, i0 - 14): i0 + 6]:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
        print("q%-3s %-22s start %9.1f end %9.1f dur %7.1f us" % (r["Queue_Id"], short(r["Kernel_Name"]), s, e, e - s))
    # E-step end -> first rounds: find the first hstep_round after i0
    j = next(k for k in range(i0, len(rows)) if "hstep_round" in rows[k]["Kernel_Name"])
    t1 = int(rows[j]["Start_Timestamp"])
    print("---- seam: E-step's last launches -> H-step's first round (t = 0 at that round's start)")
    for r in rows[j - 16: j + 4]:
        s = (int(r["Start_Timestamp"]) - t1) / 1e3; e = (int(r["End_Timestamp"]) - t1) / 1e3
        print("q%-3s %-22s start %9.1f end %9.1f dur %7.1f us" % (r["Queue_Id"], short(r["Kernel_Name"]), s, e, e - s))
PY
find $O -name "*kernel_trace.csv" -delete
