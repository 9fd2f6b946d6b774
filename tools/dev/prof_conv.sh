cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 16 32 16 32; do
O=gpurun_out/r03_conv_$v; rm -rf $O; mkdir -p $O
export PA2D_CONV_MFMA=$v
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python tools/kbench.py --engine split --only conv_fwd,conv_bwd --iters 20 > $O/kb.txt 2>&1
echo "== mfma $v"
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/stats/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(f"{int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:8.1f} us (min {float(r['MinNs'])/1e3:8.1f})  {r['Name'][:100]}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
done
