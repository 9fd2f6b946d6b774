cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $@; do
O=gpurun_out/r03_wgrad_$v; rm -rf $O; mkdir -p $O
if [ $v != base ]; then export PA2D_LIB=$GRAFT_REPO_ROOT/tools/dev/libs/lib_$v.so; else unset PA2D_LIB; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python tools/kbench.py --engine split --only conv_bwd_wonly --iters 20 > $O/kb.txt 2>&1
echo "== $v"
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/stats/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:1]:
    print(f"{int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:8.1f} us (min {float(r['MinNs'])/1e3:8.1f})  {r['Name'][:100]}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
done
