# Round-3 profile set of the dense kernels after the MFMA-shape change (run on the GPU box through gpurun):
#   bash tools/prof_r03_dense.sh [TAG]
# kernel stats of the bench per reported engine, PMC passes of the conv / linear kernels (one counter group per run,
# never combined with sys / hip traces), the MFMA-shape probe, and the zero- vs random-operand runs that show which
# kernels the chip's power management limits.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-c}
O=gpurun_out/r03_prof_$TAG
mkdir -p $O
BARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-rollout --no-folded-leg --no-exact-leg --no-darcy-leg --no-bf16-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_split --output-format csv -- python bench.py $BARGS > $O/bench_split_profiled.json 2> $O/bench_split_profiled.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_bf16s --output-format csv -- python bench.py $BARGS --engine bf16s > $O/bench_bf16s_profiled.json 2> $O/bench_bf16s_profiled.err
ONLY=conv_fwd,conv_bwd_wonly,linear_plain,linear_bias_res,linear_fwd,linear_bwd_data
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_dense_$tag --output-format csv -- python tools/kbench.py --engine split --only $ONLY --iters 3 > $O/pmc_dense_$tag.log 2>&1
done
python tools/pmc_summary.py $O/pmc_dense_* --match conv_halo > $O/pmc_conv_halo.json
python tools/pmc_summary.py $O/pmc_dense_* --match gemm_rowpanel > $O/pmc_gemm_rowpanel.json
timeout -k 5 100 tools/probes/mfma_shape_probe 1 > $O/mfma_shape_probe.txt 2>&1
timeout -k 5 100 tools/probes/mfma_shape_probe 0 >> $O/mfma_shape_probe.txt 2>&1
for z in 0 1; do
  KBENCH_ZEROS=$z timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/zeros_$z --output-format csv -- python tools/kbench.py --engine split --only conv_fwd,conv_bwd_wonly,linear_plain,slice_scatter --iters 20 > $O/zeros_$z.txt 2>&1
done
for v in 16 32; do
  PA2D_CONV_MFMA=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/convmfma_$v --output-format csv -- python tools/kbench.py --engine split --only conv_fwd --iters 20 > $O/convmfma_$v.txt 2>&1
done
timeout -k 10 200 python tools/kbench.py --engine split --iters 20 > $O/kbench_all.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
cat $O/pmc_conv_halo.json $O/mfma_shape_probe.txt; cat $O/bench_split_profiled.json | tail -1 | cut -c1-300
