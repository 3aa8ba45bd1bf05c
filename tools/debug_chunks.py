import sys, numpy as np, torch
sys.path.insert(0, '.')
from crafter_amd import BatchedEnv, state
seeds = [42, 43, 49, 51, 40, 45, 46, 55]
T = 200
tapes = []
for s in seeds:
  rs = np.random.RandomState(900 + s)
  tapes.append(rs.choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if s % 2 == 0 else rs.randint(0, 17, size=T))
tapes = np.stack(tapes, 1).astype(np.int32)
env = BatchedEnv(len(seeds), area=(256, 256), seeds=seeds, auto_reset=True)
env.reset()
dev = torch.from_numpy(tapes).to(env.device)
prev = None
for t in range(T):
  env.step(dev[t], info=False)
  torch.cuda.synchronize()
  rec = state.rec_view(env.state['rec'].cpu().numpy())
  order = env.state['chunk_order'].cpu().numpy().view(np.uint16)
  seen = env.state['chunk_seen'].cpu().numpy()
  for i in range(len(seeds)):
    n = int(rec[i]['nchunks_seen'])
    o = order[i][:n]
    if len(set(o.tolist())) != n:
      dup = [c for c in set(o.tolist()) if (o == c).sum() > 1]
      print('step', t, 'env', i, 'nchunks', n, 'dups', dup, 'episode', int(rec[i]['episode']), 'estep', int(rec[i]['step']), 'seen flag in global', [int(seen[i][c]) for c in dup], 'prev n', None if prev is None else int(prev[i]))
      sys.exit(0)
    # consistency: every chunk in order has its flag
    miss = [int(c) for c in o if not seen[i][int(c)]]
    if miss:
      po = env.state['pool_chunk_order'].cpu().numpy().view(np.uint16).reshape(2, len(seeds), -1)
      hdr = env.state['pool_hdr'].cpu().numpy()
      print('missing flags (sorted):', sorted(miss))
      print('flags set in global chunk_seen:', int(seen[i].sum()), 'of n', n, '; missing', len(miss), '; order == pool entry 0/1 order:',
            [bool((po[k][i][:n] == o).all()) for k in (0, 1)], '; flags set for old order?', 'first 8 flagged chunks', np.nonzero(seen[i])[0][:8].tolist(), 'last', np.nonzero(seen[i])[0][-8:].tolist())
      print('step', t, 'env', i, 'chunks in order without their flag in global memory:', miss[:5], 'n', n, 'episode', int(rec[i]['episode']), 'estep', int(rec[i]['step']), 'prev n', None if prev is None else int(prev[i]))
      sys.exit(0)
  prev = rec['nchunks_seen'].copy()
print('no inconsistency in', T, 'steps')
