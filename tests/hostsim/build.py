"""TEST INFRASTRUCTURE: builds tests/hostsim/_build/libhostsim.so with g++ (no GPU needed)."""
import pathlib
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
OUT = HERE / '_build' / 'libhostsim.so'
SRCS = [HERE / 'hostsim.cpp', HERE / 'wave_host.hpp'] + sorted((HERE.parent.parent / 'crafter_amd' / 'csrc').glob('*.hpp'))


def build(force=False):
  newest = max(p.stat().st_mtime for p in SRCS)
  if not force and OUT.exists() and OUT.stat().st_mtime >= newest:
    return OUT
  OUT.parent.mkdir(exist_ok=True)
  cmd = ['g++', '-std=c++17', '-O2', '-g', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared',
         '-Wall', '-Wno-unused-variable', '-Wno-unknown-pragmas', '-D__device__=', '-D__host__=',
         '-D__forceinline__=inline', '-o', str(OUT), str(HERE / 'hostsim.cpp')]
  subprocess.run(cmd, check=True)
  return OUT


if __name__ == '__main__':
  print(build(force=True))
