cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_step; rm -rf $O; mkdir -p $O
BARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-rollout --no-folded-leg --no-exact-leg --no-darcy-leg --no-bf16-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python bench.py $BARGS > $O/bench.json 2> $O/bench.err
python - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/r03_step/stats/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step', tot / 4e6, 'ms')
for r in rows[:16]:
    print(f"{float(r['TotalDurationNs'])/tot*100:6.2f}% {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
d = json.loads([l for l in open('gpurun_out/r03_step/bench.json').read().splitlines() if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
