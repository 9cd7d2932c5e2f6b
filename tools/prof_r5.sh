# usage (on the GPU box, through gpurun): bash tools/prof_r5.sh TAG [pmc]
# round-5 profile collection: the default bench line, kernel stats of bench.py under rocprofv3 (FULL bench line kept
# next to them), and with "pmc": separate FETCH_SIZE / WRITE_SIZE / SQ-counter passes (one --pmc group per run, kernel
# trace only), reduced on the box to small JSON summaries
set -x
TAG=${1:-r5a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default_with_cpu_baseline.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
if [ "$2" = "pmc" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --steps 3 --warmup 4 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --steps 3 --warmup 4 --no-cpu-baseline > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/mfma -- python bench.py --steps 3 --warmup 4 --no-cpu-baseline > $O/mfma.log 2>&1
python tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_summary.json > $O/pmc_summary.txt 2>&1
python tools/pmc_counters.py $(find $O/mfma -name "*counter_collection.csv" | head -1) $O/mfma_counters.json > $O/mfma_counters.txt 2>&1
fi
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_stats.csv" | head
cut -c1-300 $O/bench_default_with_cpu_baseline.json
du -sh $O
