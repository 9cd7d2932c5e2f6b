set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1h
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > $O/write.log 2>&1
find $O -name "*.csv" | head -20
# keep only small files
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
