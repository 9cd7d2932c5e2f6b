# usage (on the GPU box, through gpurun): bash tools/prof_r2.sh TAG [pmc]
set -x
TAG=${1:-r2a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
if [ "$2" = "pmc" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > $O/write.log 2>&1
fi
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*kernel_stats.csv" | head
cat $O/bench_default.json
du -sh $O
