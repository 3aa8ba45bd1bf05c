"""Builds crafter_amd/_lib/libcrafter_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree on purpose: the .so travels with the repo snapshot to the GPU box and is the file the
driver expects to see loaded.  ``python -m crafter_amd.build`` or __graft_entry__.build()."""
import pathlib
import shutil
import subprocess

ROOT = pathlib.Path(__file__).resolve().parent
SRC = ROOT / 'csrc' / 'crafter_hip.hip'
SRC_ROLLOUT = ROOT / 'csrc' / 'crafter_rollout.hip'   # crafter_step_n's kernels: one more flag (csrc/crafter_rollout.hpp)
OUT = ROOT / '_lib' / 'libcrafter_hip.so'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared']
CFLAGS = [f for f in FLAGS if f != '-shared']
ROLLOUT_FLAGS = ['-mllvm', '-disable-machine-licm']


def sources():
  return ([SRC, SRC_ROLLOUT] + sorted((ROOT / 'csrc').glob('*.hpp')) + sorted((ROOT / 'csrc').glob('*.inc')) +
          [ROOT.parent / 'include' / 'crafter_hip.h', ROOT.parent / 'include' / 'crafter_hip_types.h'])


def source_hash():
  """sha256 (16 hex digits) over the kernel sources: profiles quote it so that a counter measurement can be matched
  with the code it was taken from (bench.py refuses to quote HBM traffic measured on other sources)."""
  import hashlib
  h = hashlib.sha256()
  for p in sorted(sources(), key=lambda q: q.name):
    h.update(p.name.encode() + b'\0' + p.read_bytes() + b'\0')
  return h.hexdigest()[:16]


def is_stale():
  return (not OUT.exists()) or OUT.stat().st_mtime < max(p.stat().st_mtime for p in sources())


def build(force=False, verbose=False, out=None, defines=(), root=None):
  """out / defines / root: another output file, extra -D flags, another source tree (A/B builds: tools/ab_make.sh)."""
  if out is None and not force and not is_stale():
    return OUT
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  target = pathlib.Path(out) if out else OUT
  target.parent.mkdir(exist_ok=True)
  csrc = ROOT / 'csrc' if root is None else pathlib.Path(root) / 'crafter_amd' / 'csrc'
  units = [(csrc / SRC.name, []), (csrc / SRC_ROLLOUT.name, ROLLOUT_FLAGS)]
  dflags = [f'-D{d}' for d in defines]
  import tempfile
  from concurrent.futures import ThreadPoolExecutor
  with tempfile.TemporaryDirectory() as tmp:
    objs = [pathlib.Path(tmp) / (src.stem + '.o') for src, _ in units]
    cmds = [[hipcc] + CFLAGS + dflags + extra + ['-c', '-o', str(obj), str(src)] for (src, extra), obj in zip(units, objs)]
    if verbose:
      for cmd in cmds:
        print(' '.join(cmd))
    with ThreadPoolExecutor(len(cmds)) as ex:   # the units compile side by side
      procs = list(ex.map(lambda cmd: subprocess.run(cmd, capture_output=True, text=True), cmds))
    for proc in procs:
      if proc.returncode != 0:
        raise RuntimeError(f'hipcc failed:\n{proc.stdout}\n{proc.stderr}')
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', str(target)] + [str(o) for o in objs]
    if verbose:
      print(' '.join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
      raise RuntimeError(f'hipcc (link) failed:\n{proc.stdout}\n{proc.stderr}')
  return target


if __name__ == '__main__':
  print(build(force=True, verbose=True))


def resource_usage():
  """{kernel name: {'vgprs', 'scratch', 'occupancy', ...}} from -Rpass-analysis=kernel-resource-usage
  (compiles to a throw-away object; used by the tests to keep the step kernel free of scratch)."""
  import re
  import tempfile
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  text = ''
  with tempfile.TemporaryDirectory() as tmp:
    for src, extra in ((SRC, []), (SRC_ROLLOUT, ROLLOUT_FLAGS)):
      cmd = [hipcc] + CFLAGS + extra + ['-Rpass-analysis=kernel-resource-usage', '-c', '-o', str(pathlib.Path(tmp) / 'x.o'), str(src)]
      proc = subprocess.run(cmd, capture_output=True, text=True)
      if proc.returncode != 0:
        raise RuntimeError(proc.stderr)
      text += proc.stderr
  out, cur = {}, None
  for line in text.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
      name = re.search(r'crafter_[a-z_]+_kernel', m.group(1))
      targ = re.search(r'crafter_[a-z_]+_kernelILi(\d+)E(?:Li(\d+)E)?(?:Li(\d+)E)?', m.group(1))   # template instance, e.g. crafter_step_kernel<1,0>
      args = ','.join(a for a in targ.groups() if a is not None) if targ else ''
      key = (name.group(0) + (f'<{args}>' if targ else '')) if name else m.group(1)
      cur = out.setdefault(key, {})
      continue
    for key, pat in (('vgprs', r'VGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                     ('occupancy', r'Occupancy \[waves/SIMD\]: (\d+)'), ('vgpr_spill', r'VGPRs Spill: (\d+)')):
      m = re.search(pat, line)
      if m and cur is not None:
        cur[key] = int(m.group(1))
  return out
