"""TEST INFRASTRUCTURE: builds tests/c/boundary_test.c (plain C99 against include/crafter_hip.h only) and writes the
flat files it reads."""
import pathlib
import subprocess

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / 'tests' / 'c' / 'boundary_test.c'
LIB_DIR = ROOT / 'crafter_amd' / '_lib'


def compile_c(out):
  """gcc -std=c99 -pedantic -Werror: the header must be consumable by a C compiler as it stands."""
  cmd = ['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-D__HIP_PLATFORM_AMD__', f'-I{ROOT / "include"}',
         '-isystem', '/opt/rocm/include', str(SRC), '-L/opt/rocm/lib', '-lamdhip64', f'-L{LIB_DIR}', '-lcrafter_hip',
         '-Wl,-rpath,/opt/rocm/lib', f'-Wl,-rpath,{LIB_DIR}', '-o', str(out)]
  proc = subprocess.run(cmd, capture_output=True, text=True)
  assert proc.returncode == 0, proc.stderr
  return out


def write_tables(path, length=10000):
  """The host tables of crafter.Env() defaults as consecutive [int64 bytes][payload] blobs."""
  from crafter_amd import tables
  rules = tables.load_rules()
  cfg, geo = tables.make_config(1, rules, length=length)
  t = tables.HostTables(rules, tables.load_textures(), cfg, geo)
  blobs = [t.rules_bytes(), t.atlas, t.tex_tile, t.tex_icon, t.tex_digit, t.tex_alpha, t.item_pos, t.daylight, t.vignette, t.unit255]
  with open(path, 'wb') as f:
    for b in blobs:
      b = np.ascontiguousarray(b)
      f.write(np.int64(b.nbytes).tobytes())
      f.write(b.tobytes())
  return cfg
