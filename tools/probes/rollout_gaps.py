"""Summarises a rocprofv3 --kernel-trace CSV of the open loop (tools/gpu_rollout_ab.py): rollout kernel, the regeneration
kernel behind it, the gaps, and the generation batches beside them.  usage: rollout_gaps.py <dir with *kernel_trace.csv>"""
import csv, sys, pathlib, statistics as st
rows = []
for f in pathlib.Path(sys.argv[1]).rglob('*kernel_trace.csv'):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
ro = [r for r in rows if 'crafter_rollout_kernel' in r[2]]
rq = [r for r in rows if 'crafter_requeue_rollout_kernel' in r[2]]
print('rollout kernels', len(ro), 'requeue kernels', len(rq))
d1, g1, d2, g2, per = [], [], [], [], []
qi = 0
for i in range(len(ro) - 1):
  a, nxt = ro[i], ro[i + 1]
  while qi < len(rq) and rq[qi][0] < a[1]:
    qi += 1
  if qi >= len(rq) or rq[qi][0] > nxt[0]:
    continue
  q = rq[qi]
  if nxt[0] - a[0] > 3e6:   # (a pause between two rollout() calls of the driver)
    continue
  d1.append(a[1] - a[0]); g1.append(q[0] - a[1]); d2.append(q[1] - q[0]); g2.append(nxt[0] - q[1]); per.append(nxt[0] - a[0])
def show(name, v):
  v = sorted(v)
  print('%-40s median %8.2f us  mean %8.2f us  p90 %8.2f us' % (name, st.median(v) / 1e3, st.mean(v) / 1e3, v[int(0.9 * len(v))] / 1e3))
import collections
bypos = collections.defaultdict(list)
for i, g in enumerate(g2): bypos[i % 4].append(g)
for k in sorted(bypos): print('gap before the next rollout, stretch %d of a 64-step call: median %.1f us mean %.1f us' % (k, st.median(bypos[k]) / 1e3, st.mean(bypos[k]) / 1e3))
show('rollout kernel (16 steps)', d1); show('gap rollout end -> requeue start', g1); show('requeue rollout kernel', d2)
show('gap requeue end -> next rollout start', g2); show('period', per)
