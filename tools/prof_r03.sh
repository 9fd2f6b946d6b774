# Round-3 profile collection (run on the GPU box through gpurun).  Kernel stats of the bench per reported engine, then
# the PMC passes of the slice-path kernels (one counter group per run, never combined with sys/hip traces).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-a}
O=gpurun_out/r03_prof_$TAG
mkdir -p $O
BARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-rollout --no-folded-leg --no-exact-leg --no-darcy-leg --no-bf16-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_split --output-format csv -- python bench.py $BARGS > $O/bench_split_profiled.json 2> $O/bench_split_profiled.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_bf16s --output-format csv -- python bench.py $BARGS --engine bf16s > $O/bench_bf16s_profiled.json 2> $O/bench_bf16s_profiled.err
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
bash tools/prof_r03_slice.sh $TAG > $O/slice_pmc.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O/stats_split/*/ | head
