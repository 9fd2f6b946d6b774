# PMC passes of the HBM-side kernels in the forms the split engine runs (one counter group per run)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_hbm
mkdir -p $O
timeout -k 10 100 python tools/kbench.py --engine split --iters 20 --only slice_scatter,deslice,slice_bwd,slice_bwd_planes,ln_fwd,ln_fwd_planes,ln_bwd > $O/kbench_hbm.txt 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_$tag --output-format csv -- python tools/kbench.py --engine split --only slice_scatter,deslice,slice_bwd_planes,ln_fwd_planes,ln_bwd --iters 3 > $O/pmc_$tag.log 2>&1
done
python tools/pmc_summary.py $O/pmc_* --match slice > $O/pmc_slice.json
python tools/pmc_summary.py $O/pmc_* --match layernorm > $O/pmc_ln.json
grep -v amdgpu $O/kbench_hbm.txt; cat $O/pmc_slice.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
