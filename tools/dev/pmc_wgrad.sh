cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_wgrad_pmc; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES -d $O/pmc --output-format csv -- python tools/kbench.py --engine split --only conv_bwd_wonly,linear_bwd_weight --iters 3 > $O/log.txt 2>&1
python tools/pmc_summary.py $O/pmc --match gemm_mc_planes
find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
