# kernel timeline of one steady-state E-step call at C3 (two lanes): start / end / duration per launch and queue
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/estep_tl; rm -rf $O; mkdir -p $O; cd $R
OMS=${OMS:-5e-3} rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python tools/estep_rank_classes.py > $O/log.txt 2>&1
T=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if "esplit" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last E-step call: take the last 230 launches, print 40 from the middle
sel=rows[-160:-110]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    n=r["Kernel_Name"]
    short="pass_res" if "esplit_pass<5, 1" in n else "pass_w" if "esplit_pass<5, 2" in n else "lane_F" if "esplit_lane<0>" in n else "lane_M" if "esplit_lane<1>" in n else n[:30]
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print("q%-3s %-9s start %8.1f end %8.1f dur %6.1f us" % (r["Queue_Id"], short, s, e, e-s))
PY
find $O -name "*kernel_trace.csv" -delete
