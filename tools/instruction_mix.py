"""Static instruction mix of a kernel by the (inlined) source function its instructions came from: scalar ALU / vector ALU /
LDS / vector memory / scalar memory.  Answers "how much of the wave-uniform rule code already runs on the scalar unit".

  tools/kernel_sizes.sh 1 crafter_step_kernelILi1ELi1ELi1ELi0    # leaves ${TMPDIR:-/tmp}/crafter_code_size/k.s
  python tools/instruction_mix.py ${TMPDIR:-/tmp}/crafter_code_size/k.s
"""
import collections
import re
import sys

loc_re = re.compile(r'^; (/\S+):(\d+)')
fn_re = re.compile(r'^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+)?__device__[^;{]*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\(')
fn_cache = {}


def enclosing(path, ln):
  if path not in fn_cache:
    t = []
    try:
      for i, text in enumerate(open(path), 1):
        m = fn_re.match(text)
        if m:
          t.append((i, m.group(1)))
    except OSError:
      pass
    fn_cache[path] = t
  name = '?'
  for s, f in fn_cache[path]:
    if s <= ln:
      name = f
    else:
      break
  return name


def kind(op):
  if op.startswith('v_'):
    return 'valu'
  if op.startswith(('s_load', 's_buffer', 's_store')):
    return 'smem'
  if op.startswith('s_'):
    return 'salu'
  if op.startswith('ds_'):
    return 'lds'
  if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
    return 'vmem'
  return 'other'


cnt = collections.defaultdict(collections.Counter)
cur = None
for line in open(sys.argv[1]):
  m = loc_re.match(line)
  if m:
    cur = (m.group(1), int(m.group(2)))
    continue
  m = re.match(r'^\s+([a-z_0-9]+)\s', line)
  if m and cur:
    cnt[(cur[0].split('/')[-1], enclosing(*cur))][kind(m.group(1))] += 1
cols = ('salu', 'valu', 'lds', 'vmem', 'smem')
files = collections.defaultdict(collections.Counter)
print(f'{"source function":44s}' + ''.join(f'{c:>7s}' for c in cols))
for (f, fn), c in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values()))[:int(sys.argv[2]) if len(sys.argv) > 2 else 36]:
  print(f'{(f + ":" + fn):44s}' + ''.join(f'{c[k]:7d}' for k in cols))
for (f, fn), c in cnt.items():
  files[f].update(c)
print()
for f, c in sorted(files.items(), key=lambda kv: -sum(kv[1].values())):
  tot = sum(c[k] for k in cols)
  print(f'{f:44s}' + ''.join(f'{c[k]:7d}' for k in cols) + f'   scalar share of ALU {100 * c["salu"] / max(1, c["salu"] + c["valu"]):.0f} %')
