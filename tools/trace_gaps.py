"""Launch gaps and the generator's share of kernel time from a rocprofv3 --kernel-trace directory (tools/r6_evidence.sh)."""
import csv, sys, pathlib, statistics
f = sorted(pathlib.Path(sys.argv[1]).rglob('*kernel_trace.csv'))[0]
rows = [r for r in csv.DictReader(open(f)) if 'crafter' in r['Kernel_Name']]
for r in rows:
  r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
def short(n): return n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
step = sorted([r for r in rows if 'crafter_step' in r['Kernel_Name']], key=lambda r: r['s'])
req = sorted([r for r in rows if 'requeue_reset' in r['Kernel_Name']], key=lambda r: r['s'])
gen = [r for r in rows if 'crafter_gen_' in r['Kernel_Name']]
st = step[len(step) // 2:]   # the steady state: the second half of the run
t0, t1 = st[0]['s'], st[-1]['e']
per = [(b['s'] - a['s']) / 1e3 for a, b in zip(st, st[1:])]
dur = [(r['e'] - r['s']) / 1e3 for r in st]
print('rocprofv3 --kernel-trace of bench.py (4096 envs, closed loop), second half of the run: %d step launches (%s)' % (len(st), short(st[0]['Kernel_Name'])))
print('step kernel            median %7.2f us  mean %7.2f us' % (statistics.median(dur), statistics.mean(dur)))
rq = [(r['e'] - r['s']) / 1e3 for r in req if t0 <= r['s'] <= t1]
print('requeue kernel         median %7.2f us  mean %7.2f us' % (statistics.median(rq), statistics.mean(rq)))
print('step period            median %7.2f us  mean %7.2f us' % (statistics.median(per), statistics.mean(per)))
tot = {}
for r in gen:
  if t0 <= r['s'] <= t1:
    tot.setdefault(short(r['Kernel_Name']), []).append((r['e'] - r['s']) / 1e3)
gsum = sum(sum(v) for v in tot.values())
ssum = sum(dur) + sum(rq)
for k, v in sorted(tot.items()):
  print('%-36s %4d launches  mean %8.1f us  max %8.1f us  sd %7.1f us' % (k, len(v), statistics.mean(v), max(v), statistics.pstdev(v)))
print('generation kernels: %.1f us of (low-priority, side-stream) kernel time per step period (round 5: 74.0); share of all crafter kernel time in the '
      'window: %.1f %% (round 5: 49 %%)' % (gsum / max(len(per), 1), 100 * gsum / (gsum + ssum)))
