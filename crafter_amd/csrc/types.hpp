// Plain-data types shared by the gfx950 kernels, the C ABI (include/crafter_hip.h mirrors the
// ABI-visible ones field for field) and the host-side table builder (crafter_amd/tables.py).
// Only fixed-width integers, doubles and pointers: these structs are filled through ctypes.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace crafter {

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int CHUNK = 12;        // reference env.py:40 chunk size (12, 12)
constexpr int MAX_ITEMS = 16;    // data.yaml items (16 in the reference)
constexpr int MAX_ACH = 32;      // data.yaml achievements (22 in the reference)
constexpr int MAX_MATERIALS = 16;
constexpr int MAX_ACTIONS = 32;
constexpr int MAX_PLACE = 8;
constexpr int MAX_MAKE = 8;
constexpr int MAX_USES = 4;

// Object classes in the order of the reference's SemanticView list (env.py:47-49), so the
// semantic id of an object is n_materials + type.
enum : uint8_t { T_NONE = 0, T_PLAYER = 1, T_COW = 2, T_ZOMBIE = 3, T_SKELETON = 4, T_ARROW = 5, T_PLANT = 6 };

// action kinds (data.yaml action names are decoded on the host, objects.py:109-123)
enum : uint8_t { A_NOOP = 0, A_MOVE = 1, A_DO = 2, A_SLEEP = 3, A_PLACE = 4, A_MAKE = 5 };

// status bits (sticky, per env): the product fails loudly on any of these
enum : uint32_t {
  ST_OBJ_OVERFLOW = 1u,    // object table capacity exceeded
  ST_BAD_ACTION = 2u,      // action index out of range (reference: IndexError, env.py:86)
  ST_STEP_OVERFLOW = 4u,   // step beyond the uploaded daylight table
  ST_CHUNK_OVERFLOW = 8u,
  ST_POOL_MISMATCH = 16u,  // a pooled world trusted by the scheduler did not hold the episode it was adopted for
  ST_PIPE_STALL = 32u,     // a bounded in-kernel wait ran out (no kernel of this build waits inside a launch: reserved; the
                           //   pipelined step kernel that set it was removed in round 5)
};

// One world object = one 16-byte record (one dwordx4 / ds_read_b128).
struct alignas(16) Obj {
  uint8_t type;     // T_*; 0 = free slot
  int8_t health;    // objects.py:25-30 (the player's health lives in the inventory instead)
  int8_t fx, fy;    // facing (player, arrow)
  uint16_t x, y;
  int32_t aux;      // zombie cooldown / skeleton reload / plant grown
  uint32_t pad;
};
static_assert(sizeof(Obj) == 16, "Obj must be 16 bytes");

struct ItemList {
  int32_t n;
  int32_t item[MAX_USES];
  int32_t amount[MAX_USES];
  int32_t ach[MAX_USES];   // for 'receive': index of achievement collect_<item>; else -1
};

struct CollectRule {       // data.yaml collect, objects.py:214-229
  int32_t valid;
  int32_t leaves;          // material id
  double probability;      // default 1
  ItemList require;
  ItemList receive;
};

struct PlaceRule {         // data.yaml place, objects.py:231-249
  int32_t valid;
  int32_t is_object;       // 1: adds a Plant; 0: sets material
  int32_t material;        // material id written for type 'material'
  int32_t ach;             // place_<name>
  uint32_t where_mask;     // bit m set: material id m allowed
  int32_t pad;
  ItemList uses;
};

struct MakeRule {          // data.yaml make, objects.py:251-261
  int32_t valid;
  int32_t item;            // produced item index
  int32_t gives;
  int32_t ach;             // make_<name>
  uint32_t nearby_mask;    // all of these materials must be in the 3x3 window
  int32_t pad;
  ItemList uses;
};

struct Rules {
  int32_t n_actions, n_materials, n_items, n_achievements;
  uint8_t action_kind[MAX_ACTIONS];
  uint8_t action_arg[MAX_ACTIONS];          // A_MOVE: dir index (left,right,up,down); A_PLACE/A_MAKE: rule index
  int32_t item_max[MAX_ITEMS];
  int32_t item_init[MAX_ITEMS];
  uint32_t walkable_mask;                   // data.yaml walkable            (objects.py:21-22)
  uint32_t player_walkable_mask;            // + lava                        (objects.py:96-97)
  uint32_t arrow_walkable_mask;             // + water, lava                 (objects.py:369-371)
  uint32_t arrow_breaks_mask;               // table, furnace                (objects.py:381)
  int32_t mat_water, mat_grass, mat_stone, mat_path, mat_sand, mat_tree, mat_lava, mat_coal,
      mat_iron, mat_diamond, mat_table, mat_furnace;
  int32_t item_health, item_food, item_drink, item_energy;
  int32_t item_wood_sword, item_stone_sword, item_iron_sword;
  int32_t ach_wake_up, ach_eat_plant, ach_defeat_zombie, ach_defeat_skeleton, ach_eat_cow;
  CollectRule collect[MAX_MATERIALS + 1];   // indexed by material id
  PlaceRule place[MAX_PLACE];
  MakeRule make[MAX_MAKE];
};

// The scalars and small tables every object update reads (walkable masks, material / item ids, ...): the part of
// Rules in front of the collect / place / make tables.  The kernels keep a copy of it in LDS: through the global
// pointer each access is a vector load from memory (~700 clk on the rule code's critical path).
#define CRAFTER_RULES_HEAD_BYTES ((int)((offsetof(crafter::Rules, collect) + 15) / 16 * 16))

// The compiled rules of the baked data.yaml as constants (tools/bake_default_rules.py): the step kernel has an
// instance that reads its rules from here, so that material / item ids, masks and limits fold into the code.
struct alignas(8) DefaultRulesWords {
  uint32_t w[sizeof(Rules) / 4];
};
static constexpr DefaultRulesWords kDefaultRules = {{
#include "default_rules.inc"
}};
// action_kind / action_arg of the default rules as two packed literals, 3 bits per action: the one table the rule code
// indexes with a run-time value on EVERY step -- through kDefaultRules that is a load from global memory (plus a wait
// for everything else in flight) at the head of the serial rule phase; from a literal it is a scalar shift and mask.
constexpr int kPackedActions = 21;   // 63 bits
__host__ __device__ constexpr uint32_t default_rules_byte(int off) { return (kDefaultRules.w[off / 4] >> (8 * (off % 4))) & 0xFFu; }
__host__ __device__ constexpr bool default_actions_pack(int field_off) {
  for (int a = 0; a < MAX_ACTIONS; a++)
    if (default_rules_byte(field_off + a) >= (a < kPackedActions ? 8u : 1u)) return false;
  return true;
}
__host__ __device__ constexpr uint64_t default_actions_packed(int field_off) {
  uint64_t v = 0;
  for (int a = 0; a < kPackedActions; a++) v |= (uint64_t)default_rules_byte(field_off + a) << (3 * a);
  return v;
}
static_assert(default_actions_pack(offsetof(Rules, action_kind)) && default_actions_pack(offsetof(Rules, action_arg)),
              "default action table does not fit 3 bits x 21 actions");
// item_max of the default rules, 4 bits per item: the per-step inventory clamp (objects.py:126-128) reads it per lane
__host__ __device__ constexpr uint32_t default_rules_word(int off) { return kDefaultRules.w[off / 4]; }
__host__ __device__ constexpr bool default_item_max_packs() {
  for (int i = 0; i < MAX_ITEMS; i++)
    if (default_rules_word(offsetof(Rules, item_max) + 4 * i) >= 16u) return false;
  return MAX_ITEMS <= 16;
}
__host__ __device__ constexpr uint64_t default_item_max_packed() {
  uint64_t v = 0;
  for (int i = 0; i < MAX_ITEMS && i < 16; i++) v |= (uint64_t)default_rules_word(offsetof(Rules, item_max) + 4 * i) << (4 * i);
  return v;
}
static_assert(default_item_max_packs(), "default item limits do not fit 4 bits x 16 items");
constexpr uint64_t kDefaultItemMax = default_item_max_packed();
constexpr uint64_t kDefaultActionKinds = default_actions_packed(offsetof(Rules, action_kind));
constexpr uint64_t kDefaultActionArgs = default_actions_packed(offsetof(Rules, action_arg));

// Static configuration of one batch of environments (reference Env.__init__, env.py:27-56).
struct Config {
  int32_t num_envs;
  int32_t W, H;               // area
  int32_t view_w, view_h;     // view (9, 9)
  int32_t size_w, size_h;     // obs size (64, 64)
  int32_t unit_x, unit_y;     // size // view
  int32_t local_gw, local_gh; // LocalView grid (9, 7)
  int32_t item_gw, item_gh;   // ItemView grid (9, 2)
  int32_t border_x, border_y; // env.py:127
  int32_t icon_w, icon_h;     // int(0.8 * unit)  engine.py:239
  int32_t digit_w, digit_h;   // int(0.6 * unit)  engine.py:246
  int32_t max_objects;        // capacity C of the object table (slot 0 reserved)
  int32_t nchunk_x, nchunk_y; // ceil(W / 12), ceil(H / 12)
  int32_t length;             // 0 = None
  int32_t update_dist;        // 2 * max(view)   env.py:88
  int32_t n_daylight;         // entries in the daylight table
  int32_t auto_reset;         // 1: a done env is regenerated inside step()
  int32_t want_semantic;      // 1: write info['semantic'] every step
  int32_t render_obs;         // 0: skip pixels (the night RNG draw still happens)
  int32_t reward;             // 0: returned reward is forced to 0.0 (env.py:116-117)
  int32_t step_threads;       // 0 or the build's fixed step / render workgroup size (256)
  int32_t reset_threads;      // 0 or the build's fixed reset / generation workgroup size (1024)
  int32_t gen_period;         // world pool: steps between generation batches (0 = default 8, < 0 = pool off)
};

// Per-env scalar record kept in HBM between launches.
struct alignas(16) EnvRec {
  int32_t mt_pos;             // MT19937 index, 624 = twist before next draw
  int32_t step;               // Env._step
  int32_t episode;            // Env._episode
  int32_t nobj;               // slots in use incl. reserved slot 0 (next free slot)
  uint64_t seed_lane;         // CPython hash(seed) as an unsigned 64-bit lane (env.py:74)
  int32_t nchunks_seen;
  uint32_t status;            // ST_* bits, sticky
  int32_t inv[MAX_ITEMS];
  int32_t ach[MAX_ACH];
  int32_t hunger2, thirst2, fatigue2, recover2;  // 2x fixed point of objects.py:79-82
  int32_t player_last_health; // Player._last_health (objects.py:78)
  int32_t env_last_health;    // Env._last_health    (env.py:77)
  uint32_t unlocked;          // bitmask over achievements (Env._unlocked)
  int32_t sleeping;
  // outputs of the latest step (so the N=1 facade can rebuild exact Python floats)
  int32_t dhealth;            // health - last_health (reward numerator, env.py:97)
  uint32_t new_unlocked;      // achievements unlocked by the latest step
  int32_t dead;
  int32_t done;
  int32_t needs_reset;        // set by step when auto_reset and done
  // running totals of the episode (so a finished episode can be reported after an auto-reset)
  int32_t ep_dhealth;         // sum of dhealth over the episode's steps
  int32_t ep_unlock_steps;    // number of steps that unlocked something (+1.0 reward each, env.py:102-104)
  int32_t pad[1];
};

// Header of one pre-generated world (the world pool, see env_kernels.hpp gen_body / adopt_world).
struct alignas(16) PoolHdr {
  uint64_t ready;        // (generation batch sequence << 32) | episode the entry holds; one 8-byte store
  int32_t mt_pos;
  int32_t nobj;
  int32_t nchunks_seen;
  int32_t pad;
  int32_t pending;       // episode whose generation into this entry has been requested and is not through its batch yet (0: none):
                         //   a second writer of the entry must wait for it (request_generation defers)
  int32_t pad2;
};
static_assert(sizeof(PoolHdr) == 32, "PoolHdr must be 32 bytes");

// Caller-owned device buffers (torch tensors); the library never allocates or frees these.
struct StatePtrs {
  uint8_t* mat;          // [N][W*H]        material ids, index x*H + y (reference _mat_map[x][y])
  uint16_t* objmap;      // [N][W*H]        slot id per cell, 0 = empty  (reference _obj_map)
  Obj* objs;             // [N][C]          slot table, slot 0 unused, slot 1 = player
  uint32_t* mt;          // [N][624]        MT19937 key
  EnvRec* rec;           // [N]
  uint16_t* chunk_order; // [N][nchunks]    chunk ids in first-touch order (engine.py:36 dict order)
  uint8_t* chunk_seen;   // [N][nchunks]
  int32_t* census;       // [N][nchunks][5] per chunk: grass cells, path cells (kept current on every material
                         //   write), zombies, skeletons, cows (recounted by each balance pass)
  uint8_t* semantic;     // [N][W*H] or null
  uint64_t* prof;        // [N][16] shader-clock stamps: step kernel phases [0..7], reset kernel [8..15]; or null
  int32_t* reset_q;      // [2][N + 4] per step parity: count (+3 pad) then env ids that must be regenerated
  // world pool: upcoming worlds of every env, generated ahead of time on side streams.  Two entries per
  // env, indexed by episode parity, so that generations of consecutive episodes (which may run
  // concurrently on different streams) never write the same entry.
  uint8_t* pool_mat;          // [2][N][W*H]
  Obj* pool_objs;             // [2][N][C]
  uint32_t* pool_mt;          // [2][N][624]  RandomState key right after worldgen
  PoolHdr* pool_hdr;          // [2][N]
  uint16_t* pool_chunk_order; // [2][N][nchunks]
  int32_t* gen_q;             // [8][4N + 4] ring of request segments: count (+3 pad) then up to 2N (env, episode) pairs
  int32_t* gen_latest;        // [N] episode of the newest generation request of each env
  // what a stats recorder needs of an episode that just ended (recorder.py:53-66), written at done
  int32_t* terminal;          // [N][MAX_ACH + 4]: achievements[MAX_ACH], length, sum dhealth, unlock steps, episode; or null
  int32_t* pool_stats;        // [4] counters since bind: worlds adopted from the pool, envs regenerated inline although the pool
                              //   is on (world not ready in time), -, -; or null
  uint8_t* pool_perm;         // [2][N][512] OpenSimplex perm[256] | pg3[256] of the world being generated (hand-off between
                              //   the seeding and the classification kernels)
  int32_t* pool_census;       // [2][N][nchunks][5] the pooled world's grass / path cell counts per chunk (the creature counts are 0):
                              //   counted by the generator, so that adopting a world is a copy and not a pass over its map
};

// Library-owned read-only tables (uploaded once per handle).
struct TablePtrs {
  const Rules* rules;
  const uint8_t* atlas;        // RGBA texels, [x][y] order per texture
  const int32_t* tex_tile;     // [n_tex]  byte offset of each unit-sized texture, -1 if absent
  const int32_t* tex_icon;     // [MAX_ITEMS] byte offset of each item icon
  const int32_t* tex_digit;    // [11] digits '1'..'9' at [1..9], 'unknown' at [10]
  const uint8_t* tex_alpha;    // [n_tex + MAX_ITEMS + 11] 1 if the source PNG had an alpha channel
  const int32_t* item_pos;     // [MAX_ITEMS][4] icon x,y and digit x,y inside the item view
  const double* daylight;      // [n_daylight] env.py:135-139 evaluated by numpy on the host
  const double* vignette;      // [local_w][local_h] engine.py:213-218 evaluated by numpy
  const float* unit255;        // [256] float32(i) / float32(255)  (engine.py:279-281)
  const uint8_t* render_static;  // the renderer's static LDS block (render.hpp render_static_bytes), built once by
                                 // Renderer::build_static when the tables are uploaded
};

// texture slots inside tex_tile
enum : int32_t {
  TEX_UNKNOWN = 0,
  TEX_MATERIAL0 = 0,            // material id m -> TEX_MATERIAL0 + m (0 = None -> 'unknown')
  TEX_PLAYER_LEFT = 17, TEX_PLAYER_RIGHT, TEX_PLAYER_UP, TEX_PLAYER_DOWN, TEX_PLAYER_SLEEP,
  TEX_COW, TEX_ZOMBIE, TEX_SKELETON,
  TEX_ARROW_LEFT, TEX_ARROW_RIGHT, TEX_ARROW_UP, TEX_ARROW_DOWN,
  TEX_PLANT, TEX_PLANT_RIPE,
  TEX_COUNT
};

}  // namespace crafter
