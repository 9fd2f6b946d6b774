set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_prof
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_stages.py tests/test_gpu_bf16_storage.py -m gpu -q -k "slice or bf16" > $O/slice_tests.log 2>&1; tail -3 $O/slice_tests.log
(for m in bf f32; do echo "== mfma=$m"; PA2D_SLICE_MFMA=$m timeout -k 10 100 python tools/kbench.py --engine split --iters 30 --only slice_scatter,deslice; done) > $O/kbench_slice.txt 2>&1; grep -v amdgpu $O/kbench_slice.txt
BARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-rollout --no-folded-leg --no-exact-leg --no-darcy-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_split --output-format csv -- python bench.py $BARGS > $O/bench_split_profiled.json 2> $O/bench_split_profiled.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_bf16s --output-format csv -- python bench.py $BARGS --engine bf16s > $O/bench_bf16s_profiled.json 2> $O/bench_bf16s_profiled.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_conv_$tag --output-format csv -- python tools/kbench.py --engine split --only conv_fwd,conv_bwd_wonly --iters 3 > $O/pmc_conv_$tag.log 2>&1
done
python tools/pmc_summary.py $O/pmc_conv_* --match conv_halo > $O/pmc_conv_halo.json
python tools/pmc_summary.py $O/pmc_conv_* --match gemm_mc_planes_big > $O/pmc_mc_planes_big.json
cat $O/pmc_conv_halo.json $O/pmc_mc_planes_big.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
ls $O/stats_split/*/ | head
