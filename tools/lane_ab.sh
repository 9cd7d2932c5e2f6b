# lane-per-task (estep_lane.h) against wave-per-task E-step kernels at C3: wall per E-step call by omega, then kernel stats
OMS=${OMS:-2e-3,5e-3,8e-3}
for P in 0 1; do echo "VLGP_ESTEP_LANEPT=$P"; VLGP_ESTEP_LANEPT=$P OMS=$OMS python tools/estep_rank_classes.py 2>&1 | grep omega; done
echo "one stream"; VLGP_ESTEP_LANEPT=1 VLGP_ESTEP_LANES=1 OMS=$OMS python tools/estep_rank_classes.py 2>&1 | grep omega
echo "clocks, one stream"; VLGP_ESTEP_LANEPT=1 VLGP_ESTEP_LANES=1 OMS=$OMS python tools/lane_clock.py 2>&1 | grep omega
echo "clocks, two streams"; VLGP_ESTEP_LANEPT=1 OMS=$OMS python tools/lane_clock.py 2>&1 | grep omega
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lane_ab; rm -rf $O; mkdir -p $O; cd $R
for LN in 1 2; do
VLGP_ESTEP_LANEPT=1 VLGP_ESTEP_LANES=$LN OMS=5e-3 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$LN -- python tools/estep_rank_classes.py > $O/log$LN.txt 2>&1
S=$(find $O/stats$LN -name "*kernel_stats.csv" | head -1)
echo "streams: $LN"
python - "$S" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    if "esplit_lane" in r["Name"] or "esplit_pass" in r["Name"]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg %8.1f us" % (float(r["AverageNs"])/1e3), "tot %8.2f ms" % (float(r["TotalDurationNs"])/1e6))
PY
done
find $O -name "*kernel_trace.csv" -delete
