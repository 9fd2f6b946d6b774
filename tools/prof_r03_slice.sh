# Round-3 PMC passes of the slice-path kernels (one counter group per run; never combined with sys/hip traces).
# usage (on the GPU box through gpurun): bash tools/prof_r03_slice.sh [TAG]
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_slice_${1:-a}
mkdir -p $O
ONLY=slice_scatter,deslice,slice_bwd,slice_bwd_planes
timeout -k 10 100 python tools/kbench.py --engine split --iters 20 --only $ONLY > $O/kbench.txt 2>&1
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_$tag --output-format csv -- python tools/kbench.py --engine split --only $ONLY --iters 3 > $O/pmc_$tag.log 2>&1
done
python tools/pmc_summary.py $O/pmc_* --match slice > $O/pmc_slice.json
grep -v amdgpu $O/kbench.txt; cat $O/pmc_slice.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
