#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/<dir>) into the tracked summaries under profiles/.

  python tools/summarize_profile.py <tag> <stats_dir> [<fetch_dir> <write_dir> [<envs> <area> <render 0/1>]]

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table, crafter kernels first)
and, if PMC passes are given, profiles/<tag>_hbm_traffic.json with per-launch FETCH_SIZE /
WRITE_SIZE of each crafter kernel.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE
reports half the bytes of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is taken as KiB as reported.
"""
import collections
import csv
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent


def find(d, suffix):
  hits = sorted(pathlib.Path(d).rglob(f'*{suffix}'))
  if not hits:
    raise SystemExit(f'no *{suffix} under {d}')
  return hits[0]


def short(name):
  name = name.replace('(anonymous namespace)::', '')
  name = name.split('(')[0]
  return name[5:] if name.startswith('void ') else name   # template instances print as 'void name<1>(...)'


def main():
  tag, stats_dir = sys.argv[1], sys.argv[2]
  out = ROOT / 'profiles'
  out.mkdir(exist_ok=True)
  rows = list(csv.DictReader(open(find(stats_dir, 'kernel_stats.csv'))))
  rows.sort(key=lambda r: (not short(r['Name']).startswith('crafter'), -float(r['TotalDurationNs'])))
  with open(out / f'{tag}_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
    for r in rows:
      w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'],
                  r['MaxNs'], r['StdDev']])
  print('\n'.join(open(out / f'{tag}_kernel_stats.csv').read().split('\n')[:5]))
  if len(sys.argv) >= 5:
    res = collections.defaultdict(dict)
    for which, d in (('FETCH_SIZE', sys.argv[3]), ('WRITE_SIZE', sys.argv[4])):
      acc = collections.defaultdict(list)
      grid = {}
      for r in csv.DictReader(open(find(d, 'counter_collection.csv'))):
        if r['Counter_Name'] == which and 'crafter' in r['Kernel_Name']:
          acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
          grid[short(r['Kernel_Name'])] = (int(r['Grid_Size']), int(r['Workgroup_Size']))
      for k, v in acc.items():
        res[k][which + '_KiB_mean'] = float(np.mean(v))
        res[k][which + '_launches'] = len(v)
        res[k]['grid_threads'], res[k]['workgroup'] = grid[k]
    for k, v in res.items():
      f, wr = v.get('FETCH_SIZE_KiB_mean', 0.0), v.get('WRITE_SIZE_KiB_mean', 0.0)
      v['hbm_bytes_per_launch'] = (2.0 * f + wr) * 1024.0
      v['note'] = 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction)'
    sys.path.insert(0, str(ROOT))
    from crafter_amd.build import source_hash
    res['_source'] = {'csrc_sha16': source_hash(), 'note': 'crafter_amd.build.source_hash() of the kernel sources these counters were measured on'}
    if len(sys.argv) >= 8:   # the workload the counters were collected on: bench.py only quotes a profile of its own workload
      res['_source']['workload'] = {'envs': int(sys.argv[5]), 'area': int(sys.argv[6]), 'render': bool(int(sys.argv[7]))}
    (out / f'{tag}_hbm_traffic.json').write_text(json.dumps(res, indent=1) + '\n')
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
  main()
