set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_epmc
rm -rf $O; mkdir -p $O
cd $R
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > $O/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_epmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for key in ("estep_fast_kernel<5, 16, 16>", "mstep_accum<5, 1, 1>", "estep_long_kernel<5>"):
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
PY
