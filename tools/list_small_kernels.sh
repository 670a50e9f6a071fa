#!/bin/bash
# One train step's kernels other than the MFMA GEMMs and the Winograd transforms, in launch order with durations
# (run on the GPU box: gpurun -- bash tools/list_small_kernels.sh).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/small_kernels; mkdir -p gpurun_out/small_kernels
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/small_kernels -o t -- python3 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline --no-train3d > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/small_kernels/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step = rows after the last adam kernel before final... print the small kernels (non conv/wino) of the last step
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
tot=0
for r in rows[a+1:b+1]:
    n=r['Kernel_Name']
    if any(k in n for k in ('conv_igemm','conv_wgrad','wino_')): continue
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=d
    print(f"{d:8.1f} us grid={r['Grid_Size_X']:>10s} {n[:90]}")
print('total other us', tot)
PY
