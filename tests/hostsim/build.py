"""TEST INFRASTRUCTURE: builds tests/hostsim/_build/libhostsim.so with g++ (no GPU needed)."""
import pathlib
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
OUT = HERE / '_build' / 'libhostsim.so'
SRCS = [HERE / 'hostsim.cpp', HERE / 'wave_host.hpp'] + sorted((HERE.parent.parent / 'crafter_amd' / 'csrc').glob('*.hpp'))


def build(force=False, defines=(), tag=''):
  """defines / tag: a variant build (e.g. CRAFTER_SPRITE_ROWS=1 to force the renderer's overflow path).
  CRAFTER_LIT_SPRITE_STEPS=96 (the library: 1024): one host thread lights the sprite-row table here; the scenarios run through
  both halves -- steps whose sprite rows come lit from the table and steps that light them per frame."""
  out = OUT if not tag else OUT.with_name(f'libhostsim_{tag}.so')
  newest = max(p.stat().st_mtime for p in SRCS)
  if not force and out.exists() and out.stat().st_mtime >= newest:
    return out
  out.parent.mkdir(exist_ok=True)
  cmd = ['g++', '-std=c++17', '-O2', '-g', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared',
         '-Wall', '-Wno-unused-variable', '-Wno-unknown-pragmas', '-D__device__=', '-D__host__=',
         '-D__forceinline__=inline', '-DCRAFTER_LIT_SPRITE_STEPS=96'] + [f'-D{d}' for d in defines] + ['-o', str(out), str(HERE / 'hostsim.cpp')]
  subprocess.run(cmd, check=True)
  return out


if __name__ == '__main__':
  print(build(force=True))
