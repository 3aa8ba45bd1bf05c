// Env.reset(): world reseed, player, terrain and initial creatures -- reference env.py:70-81,
// engine.py:33-39, worldgen.py:10-91; semantics SURVEY.md A.4.
//
// The terrain noise (6.3 OpenSimplex evaluations per cell on average) has no side effects, so
// it is evaluated for all cells in parallel; what has to stay in reference order is the
// consumption of the env's MT19937 stream (worldgen.py:43,45,47,58 for materials, 71,73,75 for
// creatures).  Pass 1 therefore classifies every cell into either a final material or a "pending
// chain" code holding the noise-dependent condition bits; pass 2 walks only the pending cells in
// x-major order and draws; pass 3 does the same for creature placement.
#pragma once
#include <math.h>

#include "env_core.hpp"
#include "simplex.hpp"

namespace crafter {

// per-cell code while generating (lives in the LDS `mat` array)
enum : uint8_t {
  WG_MAT_MASK = 0x0F,
  WG_TREE = 0x10,      // pending: grassland cell with simplex(x, y, 5, 7) > 0 -> one draw
  WG_TUNNEL = 0x40,    // tunnels[x, y] (worldgen.py:40,43)
  WG_PENDING = 0x80,   // low bits = conditions c1..c4 of the mountain chain, or WG_TREE
};

template <class W>
struct WorldGen {
  Env<W>& e;
  uint8_t* perm;   // LDS [256]
  uint8_t* pg3;    // LDS [256]
  uint8_t* source; // LDS [256] scratch for the seeding shuffle
  uint8_t* ridx;   // LDS [256] shuffle indices

  __device__ WorldGen(Env<W>& env, uint8_t* lds512x2) : e(env) {
    perm = lds512x2;
    pg3 = lds512x2 + 256;
    source = lds512x2 + 512;
    ridx = lds512x2 + 768;
  }

  // OpenSimplex(seed) permutation (SURVEY App. B "Seeding"): indices in parallel, shuffle serial
  __device__ void seed_simplex(int64_t seed) {
    e.w.block_for(256, [&](int i) {
      source[i] = (uint8_t)i;
      ridx[i] = (uint8_t)simplex_shuffle_index(seed, i);
    });
    e.w.sync();
    if (e.w.wave0()) {
      for (int i = 255; i >= 0; i--) {
        int r = ridx[i];
        int v = source[r];
        e.st(perm + i, v);
        e.st(pg3 + i, (v % 24) * 3);
        e.st(source + r, source[i]);
        e.w.wsync();
      }
    }
    e.w.sync();
  }

  // worldgen.py:79-91 with a single size: 0 + 1 * noise, / 1
  __device__ static double S1(const Simplex& sx, double x, double y, double z, double size) {
    return sx.noise3(x / size, y / size, z);
  }

  // worldgen.py:21-61 up to (not including) the uniform() draws
  __device__ uint8_t classify(const Simplex& sx, int x, int y, int px, int py) const {
    const Rules& R = e.R;
    double fx = (double)x, fy = (double)y;
    int d2 = (x - px) * (x - px) + (y - py) * (y - py);
    double start = 4 - __builtin_sqrt((double)d2);
    start += 2 * S1(sx, fx, fy, 8, 3);
    start = 1 / (1 + exp(-start));
    double water = (0 + 1 * sx.noise3(fx / 15, fy / 15, 3)) + 0.15 * sx.noise3(fx / 5, fy / 5, 3);
    water = water + 0.1;
    water -= 2 * start;
    double mountain = (0 + 1 * sx.noise3(fx / 15, fy / 15, 0)) + 0.3 * sx.noise3(fx / 5, fy / 5, 0);
    mountain /= (1 + 0.3);
    mountain -= 4 * start + 0.3 * water;
    if (start > 0.5) return (uint8_t)R.mat_grass;
    if (mountain > 0.15) {
      if (S1(sx, fx, fy, 6, 7) > 0.15 && mountain > 0.3) return (uint8_t)R.mat_path;          // cave
      if (S1(sx, (double)(2 * x), fy / 5, 7, 3) > 0.4) return (uint8_t)(R.mat_path | WG_TUNNEL);  // horizontal tunnel
      if (S1(sx, fx / 5, (double)(2 * y), 7, 3) > 0.4) return (uint8_t)(R.mat_path | WG_TUNNEL);  // vertical tunnel
      int c1 = S1(sx, fx, fy, 1, 8) > 0;
      int c2 = S1(sx, fx, fy, 2, 6) > 0.4;
      int c3 = mountain > 0.18;
      int c4 = mountain > 0.3 && S1(sx, fx, fy, 6, 5) > 0.35;
      if (!(c1 | c2 | c3)) return (uint8_t)(c4 ? R.mat_lava : R.mat_stone);
      return (uint8_t)(WG_PENDING | c1 | (c2 << 1) | (c3 << 2) | (c4 << 3));
    }
    if (0.25 < water && water <= 0.35 && S1(sx, fx, fy, 4, 9) > -0.2) return (uint8_t)R.mat_sand;
    if (0.3 < water) return (uint8_t)R.mat_water;
    if (S1(sx, fx, fy, 5, 7) > 0) return (uint8_t)(WG_PENDING | WG_TREE);
    return (uint8_t)R.mat_grass;
  }

  // the draws of worldgen.py:43-50,58 for one pending cell
  __device__ int resolve(int code) {
    const Rules& R = e.R;
    if (code & WG_TREE) return (e.uniform() > 0.8) ? R.mat_tree : R.mat_grass;
    if ((code & 1) && e.uniform() > 0.85) return R.mat_coal;
    if ((code & 2) && e.uniform() > 0.75) return R.mat_iron;
    if ((code & 4) && e.uniform() > 0.994) return R.mat_diamond;
    return (code & 8) ? R.mat_lava : R.mat_stone;
  }

  // env.py:70-81
  __device__ void reset_env() {
    const Config& c = e.cfg;
    const Rules& R = e.R;
    EnvRec* rec = e.rec;
    int cells = c.W * c.H;
    int nch = c.nchunk_x * c.nchunk_y;
    int episode = rec->episode + 1;
    uint32_t wseed = world_seed(rec->seed_lane, (uint64_t)episode);
    e.w.sync();
    // World.reset engine.py:33-39
    e.w.block_for(cells, [&](int i) { e.objmap[i] = 0; e.g_objmap[i] = 0; });
    e.w.block_for(nch, [&](int i) { e.chunk_seen[i] = 0; e.chunk_order[i] = 0; });
    e.w.block_for(R.n_items, [&](int i) { rec->inv[i] = R.item_init[i]; });
    e.w.block_for(MAX_ACH, [&](int i) { rec->ach[i] = 0; });
    if (e.w.wave0()) {
      uint32_t s = wseed;  // RandomState(seed): init_genrand, serial recurrence
      for (int i = 0; i < MT_N; i++) {
        e.st(e.mt + i, s);
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1);
      }
      e.st(&rec->episode, episode);
      e.st(&rec->step, 0);
      e.st(&rec->nchunks_seen, 0);
      e.st(&rec->hunger2, 0);
      e.st(&rec->thirst2, 0);
      e.st(&rec->fatigue2, 0);
      e.st(&rec->recover2, 0);
      e.st(&rec->sleeping, 0);
      e.st(&rec->unlocked, 0);
      e.st(&rec->dhealth, 0);
      e.st(&rec->new_unlocked, 0);
      e.st(&rec->dead, 0);
      e.st(&rec->done, 0);
      e.st(&rec->needs_reset, 0);
      Obj z;
      z.type = T_NONE; z.health = 0; z.fx = 0; z.fy = 0; z.x = 0; z.y = 0; z.aux = 0; z.pad = 0;
      e.st(e.objs, z);
    }
    e.mt_pos = MT_N;
    e.nobj = 1;
    e.dirty_slots = 0;
    e.w.sync();
    int px = c.W / 2, py = c.H / 2;
    if (e.w.wave0()) {
      int h0 = rec->inv[R.item_health];
      e.st(&rec->player_last_health, h0);   // objects.py:78
      e.st(&rec->env_last_health, h0);      // env.py:77
      e.obj_add(T_PLAYER, px, py, 0, 0, 1, 0);  // facing (0, 1) objects.py:72; slot 1
    }
    e.w.sync();
    // worldgen.py:11: OpenSimplex(seed=randint(0, 2**31 - 1))
    uint32_t sseed = 0;
    if (e.w.wave0()) sseed = e.randint(2147483647u);
    sseed = e.w.bcast_from_wave0(sseed);
    seed_simplex((int64_t)sseed);
    // pass 1: classify every cell (parallel)
    Simplex sx{perm, pg3};
    e.w.block_for(cells, [&](int i) {
      int x = i / c.H, y = i - x * c.H;
      e.mat[i] = classify(sx, x, y, px, py);
    });
    e.w.sync();
    if (e.w.wave0()) {
      // pass 2: material draws in x-major order
      for (int base = 0; base < cells; base += 64) {
        uint64_t m = e.w.ballot(base, cells, [&](int i) { return (e.mat[i] & WG_PENDING) != 0; });
        while (m) {
          int b = __builtin_ctzll(m);
          m &= m - 1;
          int i = base + b;
          int mat = resolve(e.mat[i]);
          e.st(e.mat + i, mat);
        }
        e.w.wsync();
      }
      // pass 3: creatures, worldgen.py:64-76.  g/z/s = which of the three draws the cell can reach
      for (int base = 0; base < cells; base += 64) {
        uint64_t mg = 0, mz = 0, ms = 0;
        auto flags = [&](int i, int which) {
          int code = e.mat[i];
          int mat = code & WG_MAT_MASK;
          if (!((R.walkable_mask >> mat) & 1u)) return false;
          int x = i / c.H, y = i - x * c.H;
          int d2 = (x - px) * (x - px) + (y - py) * (y - py);
          if (which == 0) return d2 > 9 && mat == R.mat_grass;           // dist > 3 and grass
          if (which == 1) return d2 > 100;                                // dist > 10
          return mat == R.mat_path && (code & WG_TUNNEL) != 0;            // tunnel path
        };
        mg = e.w.ballot(base, cells, [&](int i) { return flags(i, 0); });
        mz = e.w.ballot(base, cells, [&](int i) { return flags(i, 1); });
        ms = e.w.ballot(base, cells, [&](int i) { return flags(i, 2); });
        uint64_t any = mg | mz | ms;
        while (any) {
          int b = __builtin_ctzll(any);
          uint64_t bit = 1ull << b;
          any &= any - 1;
          int i = base + b;
          int x = i / c.H, y = i - x * c.H;
          if ((mg & bit) && e.uniform() > 0.985)
            e.obj_add(T_COW, x, y, 3, 0, 0, 0);
          else if ((mz & bit) && e.uniform() > 0.993)
            e.obj_add(T_ZOMBIE, x, y, 5, 0, 0, 0);
          else if ((ms & bit) && e.uniform() > 0.95)
            e.obj_add(T_SKELETON, x, y, 3, 0, 0, 0);
        }
      }
    }
    e.w.sync();
    // strip the generation flags, publish the material map
    e.w.block_for(cells, [&](int i) {
      uint8_t m = e.mat[i] & WG_MAT_MASK;
      e.mat[i] = m;
      e.g_mat[i] = m;
    });
    e.w.sync();
  }
};

}  // namespace crafter
