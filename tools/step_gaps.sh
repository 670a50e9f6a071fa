#!/bin/bash
# Device idle time inside one train step (kernel trace of the last step between two Adam launches): busy, span, the
# largest gaps and the time per kernel.  gpurun -- bash tools/step_gaps.sh [train2d|train3d]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/gaps3d; mkdir -p gpurun_out/gaps3d
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps3d -o t -- python3 bench.py --workload ${1:-train3d} --steps 3 --warmup 2 --no-infer --no-cpu-baseline --no-train3d > gpurun_out/gaps3d/bench.json 2>/dev/null
python3 - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/gaps3d/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
seg=rows[a+1:b+1]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e3
span=(int(seg[-1]['End_Timestamp'])-int(rows[a]['End_Timestamp']))/1e3
prev=int(rows[a]['End_Timestamp']); gaps=[]
for r in seg:
    gaps.append(((int(r['Start_Timestamp'])-prev)/1e3, r['Kernel_Name'][:50])); prev=max(prev,int(r['End_Timestamp']))
print('kernels', len(seg), 'busy us', round(busy), 'span us', round(span), 'idle us', round(span-busy))
gaps.sort(reverse=True); print(gaps[:6])
import collections
c=collections.Counter(); t=collections.Counter()
for r in seg:
    n=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:40]
    c[n]+=1; t[n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for n,v in t.most_common(30): print(f"{n:42s} {c[n]:4d} {v:9.1f} us")
print(json.loads(open('gpurun_out/gaps3d/bench.json').read().strip().splitlines()[-1])['ms_per_step'])
PY
