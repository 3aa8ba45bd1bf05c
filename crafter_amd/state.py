"""Shapes and views of the per-env state buffers (types.hpp StatePtrs), backend-agnostic.

The buffers themselves are allocated by the caller (torch tensors on the GPU in the product);
this module only says how big they are and how to read them back as structured numpy arrays.
"""
import numpy as np

from . import abi


def state_spec(cfg):
  """name -> (shape, numpy dtype) of every caller-owned state buffer."""
  n, cells = cfg.num_envs, cfg.W * cfg.H
  nch = cfg.nchunk_x * cfg.nchunk_y
  return {
      'mat': ((n, cells), np.uint8),
      'objmap': ((n, cells), np.uint16),
      'objs': ((n, cfg.max_objects, abi.OBJ_DTYPE.itemsize), np.uint8),
      'mt': ((n, abi.MT_N), np.uint32),
      'rec': ((n, abi.REC_DTYPE.itemsize), np.uint8),
      'chunk_order': ((n, nch), np.uint16),
      'chunk_seen': ((n, nch), np.uint8),
      'census': ((n, nch * 5), np.int32),
      'semantic': ((n, cells), np.uint8),
      'reset_q': ((2, n + 4), np.int32),
      # world pool (only allocated with auto_reset)
      'pool_mat': ((2, n, cells), np.uint8),
      'pool_objs': ((2, n, cfg.max_objects, abi.OBJ_DTYPE.itemsize), np.uint8),
      'pool_mt': ((2, n, abi.MT_N), np.uint32),
      'pool_hdr': ((2, n, abi.POOL_HDR_DTYPE.itemsize), np.uint8),
      'pool_chunk_order': ((2, n, nch), np.uint16),
      'gen_q': ((8, 4 * n + 4), np.int32),
      'gen_latest': ((n,), np.int32),
      'terminal': ((n, abi.MAX_ACH + 4), np.int32),
      'pool_stats': ((4,), np.int32),
      'pool_perm': ((2, n, 512), np.uint8),
      'pool_census': ((2, n, nch * 5), np.int32),
  }


POOL_BUFFERS = ('pool_mat', 'pool_objs', 'pool_mt', 'pool_hdr', 'pool_chunk_order', 'gen_q', 'gen_latest', 'pool_stats', 'pool_perm', 'pool_census')


def seed_lanes(seeds):
  """CPython ``hash(seed)`` as the unsigned 64-bit lane of the tuple hash in env.py:74.
  Works for any hashable seed the reference would accept, not only ints."""
  return np.array([hash(s) & 0xFFFFFFFFFFFFFFFF for s in seeds], dtype=np.uint64)


def rec_view(rec_bytes):
  """uint8 [N, sizeof(EnvRec)] -> structured array [N]."""
  a = np.ascontiguousarray(rec_bytes)
  return a.view(abi.REC_DTYPE).reshape(a.shape[0])


def objs_view(objs_bytes):
  """uint8 [N, C, 16] -> structured array [N, C]."""
  a = np.ascontiguousarray(objs_bytes)
  return a.view(abi.OBJ_DTYPE).reshape(a.shape[0], a.shape[1])


def live_objects(objs_row, nobj, health):
  """Objects of one env in slot order as (type, x, y, health, fx, fy, aux) tuples -- the same
  canonical form as oracle.crafter_oracle.OracleEnv.objects()."""
  out = []
  for s in range(1, int(nobj)):
    o = objs_row[s]
    if o['type'] == abi.T_NONE:
      continue
    h = int(health) if o['type'] == abi.T_PLAYER else int(o['health'])
    out.append((int(o['type']), int(o['x']), int(o['y']), h, int(o['fx']), int(o['fy']), int(o['aux'])))
  return out


def occupied_cells(objs_row, nobj, cfg):
  """bool [W][H]: cells holding an object (World._obj_map != 0, engine.py:32), from the slot table."""
  occ = np.zeros((cfg.W, cfg.H), bool)
  for s in range(1, int(nobj)):
    o = objs_row[s]
    if o['type'] != abi.T_NONE:
      occ[int(o['x']), int(o['y'])] = True
  return occ


def chunk_keys(order_row, nseen, cfg):
  """chunk ids -> the reference's (xmin, xmax, ymin, ymax) keys (engine.py:112-117)."""
  keys = []
  for c in order_row[:int(nseen)]:
    cx, cy = divmod(int(c), cfg.nchunk_y)
    xmin, ymin = cx * abi.CHUNK, cy * abi.CHUNK
    keys.append((xmin, min(xmin + abi.CHUNK, cfg.W), ymin, min(ymin + abi.CHUNK, cfg.H)))
  return keys
