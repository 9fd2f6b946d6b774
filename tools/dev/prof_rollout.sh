cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_rollout; mkdir -p $O
timeout -k 10 200 python tools/dev/rollout_b1.py > $O/plain.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python tools/dev/rollout_b1.py > $O/profiled.txt 2>&1
find $O -name "*agent_info.csv" -delete
cat $O/plain.txt $O/profiled.txt | grep steps/s
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03_rollout/stats/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 20-step rollout: take the final 20*K kernels; find K = kernels between successive identical sequences
names = [r['Kernel_Name'] for r in rows]
# step boundary = the last kernel of a step (window shift copy); count kernels per step from the tail
tail = rows[-3000:]
import collections
# gaps
tot_k = sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in tail)
span = int(tail[-1]['End_Timestamp'])-int(tail[0]['Start_Timestamp'])
print('tail 3000 kernels: kernel time', tot_k/1e3, 'us  span', span/1e3, 'us  busy frac', tot_k/span)
agg = collections.defaultdict(lambda: [0,0])
for r in tail:
    a = agg[r['Kernel_Name'][:90]]; a[0]+=1; a[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,(n,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
    print(f'{t/1e3:9.1f} us {n:5d} x {t/n/1e3:7.2f} us  {k}')
gaps = [int(tail[i+1]['Start_Timestamp'])-int(tail[i]['End_Timestamp']) for i in range(len(tail)-1)]
gaps.sort()
print('gap median', gaps[len(gaps)//2], 'ns  p90', gaps[int(.9*len(gaps))], ' mean', sum(gaps)/len(gaps))
PY
find $O -name "*kernel_trace.csv" -delete
