/* libcrafter_hip.so -- the plain-data structs that cross the C ABI of crafter_hip.h, in C99.
 *
 * Everything a binding in C / Rust / Go / Java needs to fill the constructor arguments of the reference
 * (crafter.Env.__init__, env.py:27-56) and to read the state back: no C++ (the kernels' own definitions live in
 * crafter_amd/csrc/types.hpp; crafter_hip.hip static_asserts every size and field offset of this file against them,
 * so the two cannot drift), fixed-width integers, doubles and pointers only.  crafter_amd/abi.py is the ctypes
 * mirror of the same layouts.
 */
#ifndef CRAFTER_HIP_TYPES_H_
#define CRAFTER_HIP_TYPES_H_

#include <stddef.h>
#include <stdint.h>

#define CRAFTER_MT_N 624          /* words of an MT19937 key (numpy RandomState)                    */
#define CRAFTER_CHUNK 12          /* chunk edge, env.py:40                                          */
#define CRAFTER_MAX_ITEMS 16      /* data.yaml items (16 in the reference)                          */
#define CRAFTER_MAX_ACH 32        /* data.yaml achievements (22 in the reference)                   */
#define CRAFTER_MAX_MATERIALS 16
#define CRAFTER_MAX_ACTIONS 32
#define CRAFTER_MAX_PLACE 8
#define CRAFTER_MAX_MAKE 8
#define CRAFTER_MAX_USES 4

/* object classes, in the order of the reference's SemanticView list (env.py:47-49) */
enum { CRAFTER_T_NONE = 0, CRAFTER_T_PLAYER = 1, CRAFTER_T_COW = 2, CRAFTER_T_ZOMBIE = 3, CRAFTER_T_SKELETON = 4,
       CRAFTER_T_ARROW = 5, CRAFTER_T_PLANT = 6 };
/* action kinds (data.yaml action names decoded on the host, objects.py:109-123) */
enum { CRAFTER_A_NOOP = 0, CRAFTER_A_MOVE = 1, CRAFTER_A_DO = 2, CRAFTER_A_SLEEP = 3, CRAFTER_A_PLACE = 4, CRAFTER_A_MAKE = 5 };
/* sticky per-env status bits (crafter_env_rec.status) */
enum { CRAFTER_ST_OBJ_OVERFLOW = 1, CRAFTER_ST_BAD_ACTION = 2, CRAFTER_ST_STEP_OVERFLOW = 4, CRAFTER_ST_CHUNK_OVERFLOW = 8,
       CRAFTER_ST_POOL_MISMATCH = 16, CRAFTER_ST_PIPE_STALL = 32 };
/* texture slots of crafter_host_tables.tex_tile: material id m at CRAFTER_TEX_MATERIAL0 + m (0 = 'unknown'), then sprites */
enum { CRAFTER_TEX_MATERIAL0 = 0, CRAFTER_TEX_PLAYER_LEFT = 17, CRAFTER_TEX_PLAYER_RIGHT, CRAFTER_TEX_PLAYER_UP,
       CRAFTER_TEX_PLAYER_DOWN, CRAFTER_TEX_PLAYER_SLEEP, CRAFTER_TEX_COW, CRAFTER_TEX_ZOMBIE, CRAFTER_TEX_SKELETON,
       CRAFTER_TEX_ARROW_LEFT, CRAFTER_TEX_ARROW_RIGHT, CRAFTER_TEX_ARROW_UP, CRAFTER_TEX_ARROW_DOWN, CRAFTER_TEX_PLANT,
       CRAFTER_TEX_PLANT_RIPE, CRAFTER_TEX_COUNT };

#if defined(__GNUC__) || defined(__clang__)
#define CRAFTER_ALIGN16 __attribute__((aligned(16)))
#else
#define CRAFTER_ALIGN16
#endif

/* One world object (engine.py:50-57 World.add): 16 bytes. */
typedef struct crafter_obj {
  uint8_t type;      /* CRAFTER_T_*; 0 = free slot                                                  */
  int8_t health;     /* objects.py:25-30 (the player's health is inventory['health'])               */
  int8_t fx, fy;     /* facing (player, arrow)                                                      */
  uint16_t x, y;
  int32_t aux;       /* zombie cooldown / skeleton reload / plant grown                             */
  uint32_t pad;
} CRAFTER_ALIGN16 crafter_obj;

typedef struct crafter_item_list {
  int32_t n;
  int32_t item[CRAFTER_MAX_USES];
  int32_t amount[CRAFTER_MAX_USES];
  int32_t ach[CRAFTER_MAX_USES];   /* 'receive': achievement collect_<item>; else -1 */
} crafter_item_list;

typedef struct crafter_collect_rule {   /* data.yaml collect, objects.py:214-229 */
  int32_t valid;
  int32_t leaves;            /* material id written in place of the collected one */
  double probability;        /* default 1 */
  crafter_item_list require;
  crafter_item_list receive;
} crafter_collect_rule;

typedef struct crafter_place_rule {     /* data.yaml place, objects.py:231-249 */
  int32_t valid;
  int32_t is_object;         /* 1: adds a Plant; 0: sets `material` */
  int32_t material;
  int32_t ach;               /* place_<name> */
  uint32_t where_mask;       /* bit m: material id m allowed under it */
  int32_t pad;
  crafter_item_list uses;
} crafter_place_rule;

typedef struct crafter_make_rule {      /* data.yaml make, objects.py:251-261 */
  int32_t valid;
  int32_t item;              /* produced item index */
  int32_t gives;
  int32_t ach;               /* make_<name> */
  uint32_t nearby_mask;      /* all of these materials must be in the 3x3 window */
  int32_t pad;
  crafter_item_list uses;
} crafter_make_rule;

/* data.yaml compiled to integers (constants.py:6-8); material ids are 1 + position in data.yaml's list (0 = None). */
typedef struct crafter_rules {
  int32_t n_actions, n_materials, n_items, n_achievements;
  uint8_t action_kind[CRAFTER_MAX_ACTIONS];
  uint8_t action_arg[CRAFTER_MAX_ACTIONS];   /* MOVE: 0..3 = left, right, up, down; PLACE / MAKE: rule index */
  int32_t item_max[CRAFTER_MAX_ITEMS];
  int32_t item_init[CRAFTER_MAX_ITEMS];
  uint32_t walkable_mask;                    /* data.yaml walkable (objects.py:21-22) */
  uint32_t player_walkable_mask;             /* + lava            (objects.py:96-97) */
  uint32_t arrow_walkable_mask;              /* + water, lava     (objects.py:369-371) */
  uint32_t arrow_breaks_mask;                /* table, furnace    (objects.py:381) */
  int32_t mat_water, mat_grass, mat_stone, mat_path, mat_sand, mat_tree, mat_lava, mat_coal, mat_iron, mat_diamond,
      mat_table, mat_furnace;
  int32_t item_health, item_food, item_drink, item_energy;
  int32_t item_wood_sword, item_stone_sword, item_iron_sword;
  int32_t ach_wake_up, ach_eat_plant, ach_defeat_zombie, ach_defeat_skeleton, ach_eat_cow;
  crafter_collect_rule collect[CRAFTER_MAX_MATERIALS + 1];   /* indexed by material id */
  crafter_place_rule place[CRAFTER_MAX_PLACE];
  crafter_make_rule make[CRAFTER_MAX_MAKE];
} crafter_rules;

/* Static configuration of a batch: crafter.Env(area, view, size, reward, length, seed) (env.py:27-56) for num_envs
 * environments.  Derived fields exactly as the reference computes them:
 *   unit = size // view (env.py:42);  local grid = (view_w, view_h - item_rows), item_rows = ceil(n_items / view_w)
 *   (env.py:43-46);  border = (size - unit * view) // 2 (env.py:127);  icon = int(0.8 * unit), digit = int(0.6 * unit)
 *   (engine.py:239,246);  update_dist = 2 * max(view) (env.py:88);  nchunk = ceil(area / 12). */
typedef struct crafter_config {
  int32_t num_envs;
  int32_t W, H;
  int32_t view_w, view_h;
  int32_t size_w, size_h;
  int32_t unit_x, unit_y;
  int32_t local_gw, local_gh;
  int32_t item_gw, item_gh;
  int32_t border_x, border_y;
  int32_t icon_w, icon_h;
  int32_t digit_w, digit_h;
  int32_t max_objects;        /* capacity of the object table, slot 0 reserved (256 for 64x64)       */
  int32_t nchunk_x, nchunk_y;
  int32_t length;             /* 0 = None                                                            */
  int32_t update_dist;
  int32_t n_daylight;         /* entries of crafter_host_tables.daylight (length + 2)                */
  int32_t auto_reset;         /* 1: a finished env is regenerated inside crafter_step                */
  int32_t want_semantic;      /* 1: info['semantic'] written every step (state.semantic)             */
  int32_t render_obs;         /* 0: no pixels (the night noise is still drawn from the RNG)          */
  int32_t reward;             /* 0: returned reward forced to 0 (env.py:116-117)                     */
  int32_t step_threads;       /* 0 (workgroup sizes are compile-time constants of the library)       */
  int32_t reset_threads;      /* 0                                                                   */
  int32_t gen_period;         /* world pool: steps between generation batches; 0 default, < 0 off    */
} crafter_config;

/* Per-env scalar record kept in HBM between launches: what info[...] of Env.step is read from (env.py:108-115). */
typedef struct crafter_env_rec {
  int32_t mt_pos;             /* MT19937 index, 624 = twist before the next draw                     */
  int32_t step;               /* Env._step                                                           */
  int32_t episode;            /* Env._episode                                                        */
  int32_t nobj;               /* slots in use incl. reserved slot 0                                  */
  uint64_t seed_lane;         /* CPython hash(seed) as an unsigned 64-bit lane (env.py:74)           */
  int32_t nchunks_seen;
  uint32_t status;            /* CRAFTER_ST_* bits, sticky                                           */
  int32_t inv[CRAFTER_MAX_ITEMS];
  int32_t ach[CRAFTER_MAX_ACH];
  int32_t hunger2, thirst2, fatigue2, recover2;   /* 2x fixed point of objects.py:79-82              */
  int32_t player_last_health;
  int32_t env_last_health;
  uint32_t unlocked;
  int32_t sleeping;
  int32_t dhealth;            /* reward numerator of the latest step (env.py:97)                     */
  uint32_t new_unlocked;
  int32_t dead;
  int32_t done;
  int32_t needs_reset;
  int32_t ep_dhealth;
  int32_t ep_unlock_steps;
  int32_t pad[1];
} CRAFTER_ALIGN16 crafter_env_rec;

typedef struct crafter_pool_hdr {   /* header of one pre-generated world (world pool) */
  uint64_t ready;
  int32_t mt_pos;
  int32_t nobj;
  int32_t nchunks_seen;
  int32_t pad;
  int32_t pending;
  int32_t pad2;
} CRAFTER_ALIGN16 crafter_pool_hdr;

/* Caller-owned DEVICE buffers holding the world state, N = num_envs, cells = W * H, C = max_objects,
 * nch = nchunk_x * nchunk_y.  Zero-filled at allocation except rec[i].seed_lane = hash(seed_i), rec[i].mt_pos = 624,
 * rec[i].nobj = 1.  Buffers marked (pool) are only needed with auto_reset = 1 and gen_period >= 0; semantic only with
 * want_semantic; prof may be NULL.  mat / objmap / objs / mt / rec must be 16-byte aligned. */
typedef struct crafter_state_ptrs {
  uint8_t* mat;               /* [N][cells]          material ids, index x * H + y                   */
  uint16_t* objmap;           /* [N][cells]          slot per cell (scratch for LDS-resident worlds) */
  crafter_obj* objs;          /* [N][C]                                                               */
  uint32_t* mt;               /* [N][624]                                                             */
  crafter_env_rec* rec;       /* [N]                                                                  */
  uint16_t* chunk_order;      /* [N][nch]                                                             */
  uint8_t* chunk_seen;        /* [N][nch]                                                             */
  int32_t* census;            /* [N][nch][5]                                                          */
  uint8_t* semantic;          /* [N][cells] or NULL                                                   */
  uint64_t* prof;             /* [N][16] or NULL                                                      */
  int32_t* reset_q;           /* [2][N + 4]                                                           */
  uint8_t* pool_mat;          /* (pool) [2][N][cells]                                                 */
  crafter_obj* pool_objs;     /* (pool) [2][N][C]                                                     */
  uint32_t* pool_mt;          /* (pool) [2][N][624]                                                   */
  crafter_pool_hdr* pool_hdr; /* (pool) [2][N]                                                        */
  uint16_t* pool_chunk_order; /* (pool) [2][N][nch]                                                   */
  int32_t* gen_q;             /* (pool) [8][4 N + 4]                                                  */
  int32_t* gen_latest;        /* (pool) [N]                                                           */
  int32_t* terminal;          /* [N][CRAFTER_MAX_ACH + 4]: totals of the episode that just ended, or NULL */
  int32_t* pool_stats;        /* (pool) [4]                                                           */
  uint8_t* pool_perm;         /* (pool) [2][N][512]                                                   */
  int32_t* pool_census;       /* (pool) [2][N][nch][5]                                                */
} crafter_state_ptrs;

#endif /* CRAFTER_HIP_TYPES_H_ */
