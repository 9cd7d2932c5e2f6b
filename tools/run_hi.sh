python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -4
VLGP_ESTEP_SPLIT=1 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -4
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/st1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/st1/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if 'esplit_latent' in r['Name']:
            print(r['Name'].replace('(anonymous namespace)::','')[:45], r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))
PY
