"""Attributes the bytes of a kernel's machine code to the (inlined) source functions they came from.

  hipcc ... --cuda-device-only -gline-tables-only -c -o g.o crafter_hip.hip ; unbundle ; llvm-objdump -d -l
  python tools/code_size.py step.s

The step kernel has to live in a 64 KB instruction cache shared by two CUs; this is the tool that
says which helper to shrink or to stop duplicating.
"""
import collections
import re
import sys

loc_re = re.compile(r'^; (/\S+):(\d+)')
ins_re = re.compile(r'//\s*([0-9A-Fa-f]{12}):')
fn_re = re.compile(r'^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+)?__device__[^;{]*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\(')

by_line = collections.Counter()
cur = None
prev_addr = None
prev_loc = None
for line in open(sys.argv[1]):
  m = loc_re.match(line)
  if m:
    cur = (m.group(1), int(m.group(2)))
    continue
  m = ins_re.search(line)
  if m:
    addr = int(m.group(1), 16)
    if prev_addr is not None:
      by_line[prev_loc] += addr - prev_addr
    prev_addr, prev_loc = addr, cur

funcs = {}
def enclosing(path, ln):
  if path not in funcs:
    table = []
    try:
      for i, text in enumerate(open(path), 1):
        m = fn_re.match(text)
        if m:
          table.append((i, m.group(1)))
    except OSError:
      pass
    funcs[path] = table
  name = '?'
  for start, fn in funcs[path]:
    if start <= ln:
      name = fn
    else:
      break
  return name

by_fn = collections.Counter()
for (path, ln), size in by_line.items():
  by_fn[(path.split('/')[-1], enclosing(path, ln))] += size
total = sum(by_fn.values())
print('total', total)
for (f, fn), size in by_fn.most_common(45):
  print(f'{size:8d} {100 * size / total:5.1f}%  {f}:{fn}')

if len(sys.argv) > 2:
  print('--- top lines')
  for (path, ln), size in by_line.most_common(int(sys.argv[2])):
    print(f'{size:8d}  {path.split("/")[-1]}:{ln}')
