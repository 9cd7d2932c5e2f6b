# H-step alone at C3 under rocprofv3: kernel durations and gaps of its rounds (tools/hstep_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hgaps; rm -rf $O; mkdir -p $O; cd $R
python tools/hstep_gaps.py run 2>&1 | grep "H-step alone"
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python tools/hstep_gaps.py run > $O/log.txt 2>&1
T=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python tools/hstep_gaps.py reduce $T
python - "$T" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "hstep" in r["Kernel_Name"] or "mstep" in r["Kernel_Name"]]
d=collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])): print("%-62s n %5d avg %7.1f us total %8.2f ms" % (k, len(v), sum(v)/len(v), sum(v)/1e3))
PY
find $O -name "*kernel_trace.csv" -delete
