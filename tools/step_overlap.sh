#!/bin/bash
# Two-stream train step under the kernel trace: how much of a step has an MFMA kernel running, an HBM-bound
# kernel running, both, or nothing.  gpurun -- bash tools/step_overlap.sh [train2d|train3d]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"; export TMPDIR=/tmp
O="gpurun_out/overlap_${1:-train2d}"; rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- python3 bench.py --workload "${1:-train2d}" --steps 3 --warmup 2 --no-infer --no-cpu-baseline --no-train3d --no-train-e2e > "$O/bench.json" 2>/dev/null
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
f = glob.glob(O + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
MFMA = ('conv_igemm_kernel', 'conv_wgrad_kernel', 'chain64', 'wino_fused_kernel', 'wino_pre_kernel')
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# the two-stream pass is the first K steps after the warm-up: take the step between the 3rd and 4th Adam launch
if len(idx) < 4:
    sys.exit(f'the trace holds {len(idx)} Adam launches, 4 are needed (2 warm-up steps + 2 of the timed ones): did bench.py fail? see {O}/bench.json')
a, b = idx[2], idx[3]
seg = rows[a + 1:b + 1]
t0, t1 = int(rows[a]['End_Timestamp']), int(seg[-1]['End_Timestamp'])
ev = []
for r in seg:
    kind = 0 if any(k in r['Kernel_Name'] for k in MFMA) else 1
    ev.append((int(r['Start_Timestamp']), 1, kind)); ev.append((int(r['End_Timestamp']), -1, kind))
ev.sort()
n = [0, 0]; prev = t0; acc = collections.Counter()
for t, d, k in ev:
    key = ('mfma' if n[0] else '') + ('+' if n[0] and n[1] else '') + ('hbm' if n[1] else '') or 'idle'
    if n[0] >= 2 and not n[1]: key = 'mfma x2'
    if n[1] >= 2 and not n[0]: key = 'hbm x2'
    acc[key] += t - prev; prev = t; n[k] += d
span = (t1 - t0) / 1e3
print('step span us', round(span), 'kernels', len(seg), 'queues', len(set(r.get('Queue_Id', '') for r in seg)))
for k, v in acc.most_common(): print(f'  {k:10s} {v / 1e3:9.1f} us  {v / 1e3 / span:6.1%}')
c = collections.Counter(); tt = collections.Counter()
for r in seg:
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:44]
    c[n] += 1; tt[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('launches of that step (durations while sharing the device with the other stream):')
for n, v in tt.most_common(12): print(f'  {n:46s} {c[n]:4d} launches {v / c[n]:9.1f} us average')
b = json.loads(open(O + '/bench.json').read().strip().splitlines()[-1])
print('bench.py under the trace: ms_per_step', b['ms_per_step'], 'one-stream pass', b['roofline'].get('one_stream_ms_per_step'))
PY
rm -rf $O/*/ 2>/dev/null; find $O -name "*.csv" -size +20M -delete
