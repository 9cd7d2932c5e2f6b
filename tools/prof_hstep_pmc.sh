set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_hpmc_g
rm -rf $O; mkdir -p $O
cd $R
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- python tools/hstep_round_bench.py > $O/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_hpmc_g"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "hstep_round_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Grid_Size"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, {g: round(sum(x) / len(x)) for g, x in sorted(v.items(), key=lambda kv: int(kv[0]))})
PY
