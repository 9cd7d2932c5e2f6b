# usage (on the GPU box, through gpurun): bash tools/prof_r6.sh TAG [pmc]
# round-6 profile collection, ONE call = one box: the driver's bench line (CPU leg included), kernel stats of bench.py under
# rocprofv3 (the FULL bench line of that run kept next to them), the shard / C5 / cold-start lines, the two-rank
# self-launched line over the one-GPU test transport, and with "pmc": separate FETCH_SIZE / WRITE_SIZE / SQ-counter passes
# (one --pmc group per run, kernel trace only), reduced on the box to small JSON summaries.
set -x
TAG=${1:-r6a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_steps20_with_cpu_baseline.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_bench_steps10.csv
for w in C3s2 C3s4 C3s8; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_$w.json 2>/dev/null; done
python bench.py --steps 20 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_warmup1.json 2>/dev/null
python bench.py --workload C5 --steps 5 --warmup 3 --cpu-trials 6 > $O/${TAG}_bench_C5_steps5.json 2>/dev/null
python tools/estep_per_step.py > $O/${TAG}_per_iteration_from_cold.txt 2>&1
VLGP_COMM_TRANSPORT=shm VLGP_DEVICE=0 python bench.py --gpus 2 --allow-shm --workload C1 --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_selflaunch_2ranks_one_gpu_C1.json 2>/dev/null
if [ "${2:-}" = "pmc" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --steps 3 --warmup 4 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --steps 3 --warmup 4 --no-cpu-baseline > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/mfma -- python bench.py --steps 3 --warmup 4 --no-cpu-baseline > $O/mfma.log 2>&1
python tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_summary.json > $O/pmc_summary.txt 2>&1
python tools/pmc_counters.py $(find $O/mfma -name "*counter_collection.csv" | head -1) $O/${TAG}_mfma_counters.json > $O/mfma_counters.txt 2>&1
fi
rm -rf $O/stats $O/fetch $O/write $O/mfma
cut -c1-400 $O/${TAG}_bench_steps20_with_cpu_baseline.json
ls -la $O; du -sh $O
