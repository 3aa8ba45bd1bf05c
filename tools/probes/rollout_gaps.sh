cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT}
rm -rf $root/gpurun_out/rgaps_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/rgaps_trace -- python $root/tools/gpu_rollout_ab.py 4096 default > /dev/null 2>&1
python $root/tools/probes/rollout_gaps.py $root/gpurun_out/rgaps_trace
python - $root/gpurun_out/rgaps_trace <<'PY'
import csv, sys, pathlib
rows = []
for f in pathlib.Path(sys.argv[1]).rglob('*kernel_trace.csv'):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# what runs on the GPU between a requeue kernel's end and the next rollout kernel's start when the gap is large
ro = [r for r in rows if 'crafter_rollout_kernel' in r[2]]
rq = [r for r in rows if 'crafter_requeue_rollout_kernel' in r[2]]
shown = 0
for q in rq[100:]:
  nxt = [r for r in ro if r[0] > q[1]]
  if not nxt: break
  gap = nxt[0][0] - q[1]
  if 30e3 < gap < 3e6 and shown < 3:
    shown += 1
    print('gap %.1f us; kernels overlapping it:' % (gap / 1e3))
    for r in rows:
      if r[1] > q[1] and r[0] < nxt[0][0] and r is not q and r is not nxt[0]:
        print('    %-60s start %+8.1f us  end %+8.1f us (relative to the gap start)' % (r[2][:60], (r[0] - q[1]) / 1e3, (r[1] - q[1]) / 1e3))
PY
rm -rf $root/gpurun_out/rgaps_trace
