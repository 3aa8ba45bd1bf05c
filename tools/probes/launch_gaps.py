"""Summarises a rocprofv3 --kernel-trace CSV of the closed loop: the gaps between the step kernel, the regeneration kernel
behind it and the next step kernel (the `a` of T(N) = a + b N).  usage: launch_gaps.py <dir with *kernel_trace.csv>"""
import csv, sys, pathlib, statistics as st
rows = []
for f in pathlib.Path(sys.argv[1]).rglob('*kernel_trace.csv'):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r.get('Stream_Id', '')))
rows.sort()
steps = [r for r in rows if 'crafter_step_kernel' in r[2]]
req = [r for r in rows if 'crafter_requeue_reset_kernel' in r[2]]
print('step kernels', len(steps), 'requeue kernels', len(req))
g1, d2, g2, d1, per = [], [], [], [], []
ri = 0
for i in range(len(steps) - 1):
  s, nxt = steps[i], steps[i + 1]
  while ri < len(req) and req[ri][0] < s[1]:
    ri += 1
  if ri >= len(req) or req[ri][0] > nxt[0]:
    continue
  q = req[ri]
  d1.append(s[1] - s[0]); g1.append(q[0] - s[1]); d2.append(q[1] - q[0]); g2.append(nxt[0] - q[1]); per.append(nxt[0] - s[0])
def show(name, v):
  v = sorted(v)[len(v) // 10: -len(v) // 10 or None]
  print('%-38s median %7.2f us  mean %7.2f us' % (name, st.median(v) / 1e3, st.mean(v) / 1e3))
import collections
big = sorted(g2)[-len(g2) // 16:]
print('gap before the next step kernel: the largest sixteenth (the steps that launch a generation batch): median %.1f us mean %.1f us; the rest: mean %.2f us' % (
    st.median(big) / 1e3, st.mean(big) / 1e3, st.mean(sorted(g2)[:-len(g2) // 16]) / 1e3))
show('step kernel', d1); show('gap step end -> requeue start', g1); show('requeue kernel', d2); show('gap requeue end -> next step start', g2); show('step period', per)
gen = [r for r in rows if 'crafter_gen_' in r[2]]
print('generation kernels', len(gen), 'total %.1f us per step period' % (sum(r[1] - r[0] for r in gen) / 1e3 / max(1, len(steps))))
