cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_darcy; rm -rf $O; mkdir -p $O
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-rollout --no-folded-leg --no-exact-leg --no-bf16-leg"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py $BARGS > $O/bench.json 2> $O/bench.err
python - <<'PY'
import csv, glob, json, collections
f = glob.glob('gpurun_out/r03_darcy/trace/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the darcy leg runs last: take the kernels after the last NS-shaped conv... simpler: last 40 % of the trace by time
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
# find darcy region: kernels launched after the final 'adamw' of the NS part -> use grid size of conv_halo for 421x421
agg = collections.defaultdict(lambda: [0, 0])
darcy = [r for r in rows if 'conv_halo' in r['Kernel_Name']]
grids = collections.Counter(r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', '') for r in darcy)
print('conv grids', grids.most_common(4))
PY
head -1 $O/trace/*/*kernel_trace.csv | cut -c1-400
