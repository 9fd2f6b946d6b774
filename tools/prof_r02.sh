# Round-2 profile collection (run on the GPU box through gpurun): kernel-stats of the bench per reported engine, the kbench
# table, and the PMC passes (one counter group per run, never combined with sys/hip traces) of the dense kernels.
# profiles/r02_b_* came from this script as it stands; r02_c_* / r02_e_* are its two `rocprofv3 --kernel-trace --stats`
# lines re-run at later commits, followed by an un-profiled `python bench.py` on the same box (profiles/README.md has
# the exact command of every file).  A/B of one translation unit on one box: build the other variant into a second .so
# and run `PA2D_LIB=/path/to/other.so python bench.py ...` next to the default in the same gpurun call.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_prof_b
mkdir -p $O
timeout -k 10 200 python tools/kbench.py --engine split --iters 20 > $O/kbench_b32.txt 2>&1
BARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-rollout --no-folded-leg --no-exact-leg --no-darcy-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_split --output-format csv -- python bench.py $BARGS > $O/bench_split_profiled.json 2> $O/bench_split_profiled.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_bf16s --output-format csv -- python bench.py $BARGS --engine bf16s > $O/bench_bf16s_profiled.json 2> $O/bench_bf16s_profiled.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_dense_$tag --output-format csv -- python tools/kbench.py --engine split --only conv_fwd,conv_bwd_wonly,linear_plain,linear_bias_res --iters 3 > $O/pmc_dense_$tag.log 2>&1
done
python tools/pmc_summary.py $O/pmc_dense_* --match conv_halo > $O/pmc_conv_halo.json
python tools/pmc_summary.py $O/pmc_dense_* --match gemm_mc_planes_big > $O/pmc_mc_planes_big.json
python tools/pmc_summary.py $O/pmc_dense_* --match gemm_panel > $O/pmc_gemm_panel.json
cat $O/pmc_conv_halo.json $O/pmc_mc_planes_big.json $O/pmc_gemm_panel.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
ls $O/stats_split/*/ | head
