# usage (GPU box): bash tools/prof_pmc_kernels.sh TAG "<command>"
# SQ / cache counter groups (one --pmc group per run, kernel trace only) over <command>, reduced to per-kernel means
set -x
TAG=${1:-k}
CMD=${2:-"python tools/pass_quant.py"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd $R
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/g$i -- $CMD > $O/g$i.log 2>&1
done
TAG_=$TAG python - <<'PY'
import csv, glob, collections, os, json
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_" + os.environ.get("TAG_", "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: dict({c: sum(x) / len(x) for c, x in v.items()}, launches=max(len(x) for x in v.values())) for k, v in acc.items()}
json.dump(out, open(root + "/summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    if v["launches"] >= 3:
        print(k, {c: round(x) for c, x in sorted(v.items())})
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
