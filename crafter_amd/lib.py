"""ctypes binding of libcrafter_hip.so (include/crafter_hip.h).  There is no CPU path: if the
library is missing, cannot be loaded, or reports a different struct layout, import fails loudly."""
import ctypes as C
import pathlib

from . import abi

LIB_PATH = pathlib.Path(__file__).resolve().parent / '_lib' / 'libcrafter_hip.so'

EXPORTS = [
    'crafter_struct_sizes', 'crafter_abi_version', 'crafter_create', 'crafter_destroy',
    'crafter_upload_tables', 'crafter_bind_state', 'crafter_lds_bytes', 'crafter_slot_map_derived', 'crafter_step_instance', 'crafter_reset', 'crafter_step', 'crafter_step_n', 'crafter_debug_dispatch_order', 'crafter_debug_set_dispatch_order',
    'crafter_render', 'crafter_set_timing', 'crafter_get_timing', 'crafter_pool_status', 'crafter_pool_error',
    'crafter_last_error', 'crafter_debug_eval', 'crafter_extend_daylight',
    'crafter_exchange_unique_id', 'crafter_exchange_create', 'crafter_exchange_destroy', 'crafter_step_exchange',
    'crafter_exchange_wait', 'crafter_exchange_error',
]


class HostTablesC(C.Structure):
  _fields_ = [
      ('rules', C.c_void_p),
      ('atlas', C.c_void_p), ('atlas_bytes', C.c_size_t),
      ('tex_tile', C.c_void_p), ('n_tex_tile', C.c_int32),
      ('tex_icon', C.c_void_p), ('n_tex_icon', C.c_int32),
      ('tex_digit', C.c_void_p), ('n_tex_digit', C.c_int32),
      ('tex_alpha', C.c_void_p), ('n_tex_alpha', C.c_int32),
      ('item_pos', C.c_void_p), ('n_item_pos', C.c_int32),
      ('daylight', C.c_void_p), ('n_daylight', C.c_int32),
      ('vignette', C.c_void_p), ('n_vignette', C.c_int32),
      ('unit255', C.c_void_p), ('n_unit255', C.c_int32),
  ]


class CrafterLibError(RuntimeError):
  pass


_lib = None


def load(path=None):
  """Loads (once) and type-annotates the shared library.  Raises CrafterLibError if it is absent:
  build it with ``python -m crafter_amd.build`` (hipcc, gfx950)."""
  global _lib
  if _lib is not None:
    return _lib
  import os
  path = pathlib.Path(path or os.environ.get('CRAFTER_HIP_LIB') or LIB_PATH)   # env override: A/B builds
  if not path.exists():
    raise CrafterLibError(
        f'{path} not found: the HIP extension is required (no CPU fallback). '
        'Build it with `python -m crafter_amd.build`.')
  try:
    lib = C.CDLL(str(path))
  except OSError as e:
    raise CrafterLibError(f'cannot load {path}: {e}') from e
  missing = [n for n in EXPORTS if not hasattr(lib, n)]
  if os.environ.get('CRAFTER_HIP_LIB'):   # an A/B build of an older ABI (tools/ab_make.sh): everything but the newer entry points works
    missing = [n for n in missing if n != 'crafter_extend_daylight' and not n.startswith(('crafter_exchange', 'crafter_step_exchange'))]
  if missing:
    raise CrafterLibError(f'{path} lacks symbols {missing}')
  vp, i32 = C.c_void_p, C.c_int32
  lib.crafter_struct_sizes.argtypes = [C.POINTER(i32)]
  lib.crafter_struct_sizes.restype = None
  lib.crafter_abi_version.restype = i32
  lib.crafter_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
  lib.crafter_destroy.argtypes = [vp]
  lib.crafter_destroy.restype = None
  lib.crafter_upload_tables.argtypes = [vp, C.POINTER(HostTablesC)]
  lib.crafter_bind_state.argtypes = [vp, C.POINTER(abi.StatePtrs)]
  if hasattr(lib, 'crafter_extend_daylight'):
    lib.crafter_extend_daylight.argtypes = [vp, vp, i32]
  lib.crafter_lds_bytes.argtypes = [vp]
  lib.crafter_lds_bytes.restype = i32
  lib.crafter_slot_map_derived.argtypes = [vp]
  lib.crafter_slot_map_derived.restype = i32
  lib.crafter_step_instance.argtypes = [vp]
  lib.crafter_step_instance.restype = i32
  lib.crafter_reset.argtypes = [vp, vp, vp, vp]
  lib.crafter_step.argtypes = [vp, vp, vp, vp, vp, vp]
  lib.crafter_step_n.argtypes = [vp, i32, vp, vp, vp, vp, vp]
  lib.crafter_debug_dispatch_order.argtypes = [vp, vp]
  lib.crafter_debug_set_dispatch_order.argtypes = [vp, vp]
  lib.crafter_render.argtypes = [vp, vp, vp, vp]
  lib.crafter_set_timing.argtypes = [vp, i32]
  lib.crafter_get_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)]
  lib.crafter_pool_status.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
  lib.crafter_pool_error.argtypes = [vp]
  lib.crafter_pool_error.restype = C.c_char_p
  lib.crafter_debug_eval.argtypes = [i32, vp, vp, vp, vp, vp, C.c_int64, vp]
  lib.crafter_last_error.argtypes = [vp]
  lib.crafter_last_error.restype = C.c_char_p
  if hasattr(lib, 'crafter_step_exchange'):
    lib.crafter_exchange_unique_id.argtypes = [vp]
    lib.crafter_exchange_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    lib.crafter_exchange_destroy.argtypes = [vp]
    lib.crafter_exchange_destroy.restype = None
    lib.crafter_step_exchange.argtypes = [vp, vp, i32, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, i32, vp]
    lib.crafter_exchange_wait.argtypes = [vp, i32, vp]
    lib.crafter_exchange_error.argtypes = [vp]
    lib.crafter_exchange_error.restype = C.c_char_p
  sizes = (i32 * 6)()
  lib.crafter_struct_sizes(sizes)
  abi.check_sizes(list(sizes))
  _lib = lib
  return lib


def last_error(lib, handle):
  msg = lib.crafter_last_error(handle)
  return msg.decode() if msg else 'unknown error'
