/*
 * oc_amd.h — C-ABI of liboc_amd.so, the MI355X (gfx950) batched Overcooked hot path.
 *
 * The reference (HumanCompatibleAI/overcooked_ai) is pure Python and has no FFI; the
 * boundary this library sits under is the Python class API
 *     OvercookedGridworld.get_state_transition   src/overcooked_ai_py/mdp/overcooked_mdp.py:1375
 *     OvercookedGridworld.lossless_state_encoding src/overcooked_ai_py/mdp/overcooked_mdp.py:2385
 *     OvercookedEnv.step / reset / is_done        src/overcooked_ai_py/mdp/overcooked_env.py:244,288,321
 * Each entry point below names the reference function(s) it replaces.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer whose name starts with d_ is a DEVICE pointer (HBM) owned by the caller;
 *    the library never allocates, frees or synchronises.
 *  - all launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *  - return value: 0 on success, negative OC_E* on error; oc_last_error() gives a message.
 *  - thread-safe per (device, stream); the library keeps no global mutable state except the
 *    thread-local error string.
 *
 * Wire format of one environment ("env") — see DESIGN.md §3.
 *  State is struct-of-arrays over 16-byte words ("planes"): plane p of env e lives at
 *      ((oc_u128*)d_state)[p * n_envs + e]
 *  so a wavefront reading plane p for 64 consecutive envs issues one 1 KiB coalesced load.
 *  Plane 0 (header), byte offsets:
 *      0  player0 cell index (y*W + x)      3  player1 cell index (0xFF = absent, 1-player layouts)
 *      1  player0 orientation (0..3)        4  player1 orientation
 *      2  player0 held object code          5  player1 held object code
 *      6..7  timestep, u16 little endian
 *      8..15 pot slot k: cooking_tick + 1   (0 = idle, i.e. reference _cooking_tick == -1)
 *  Planes 1..n_obj_planes: one object code per grid cell; cell c is byte (c & 15) of plane 1 + (c >> 4).
 *  Object code: 0 none, 1 onion, 2 tomato, 3 dish,
 *               0x80 | (n_ingredients << 3) | tomato_bits   for a soup, where bit i of tomato_bits
 *               says ingredient i (insertion order, SoupState._ingredients, mdp.py:453) is a tomato.
 *  A soup that is held or lies on a counter is always cooked (it left its pot through the
 *  dish pickup at mdp.py:1525-1539); its tick is implied (= the recipe's cook time).
 */
#ifndef OC_AMD_H
#define OC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OC_ABI_VERSION 6

#define OC_MAX_CELLS 128
#define OC_MAX_POTS 8
#define OC_NUM_LAYERS 26 /* lossless_state_encoding layers, mdp.py:2393-2442 */
#define OC_NUM_ACTIONS 6

/* terrain codes = the reference's own TYPE_TO_CODE table (layout_generator.py:18-27) */
#define OC_T_FLOOR 0
#define OC_T_COUNTER 1
#define OC_T_ONION_DISP 2
#define OC_T_TOMATO_DISP 3
#define OC_T_POT 4
#define OC_T_DISH_DISP 5
#define OC_T_SERVE 6

/* action indices = Action.INDEX_TO_ACTION (actions.py:49-52); orientation = Direction index (actions.py:16) */
#define OC_A_NORTH 0
#define OC_A_SOUTH 1
#define OC_A_EAST 2
#define OC_A_WEST 3
#define OC_A_STAY 4
#define OC_A_INTERACT 5

/* object codes */
#define OC_O_NONE 0
#define OC_O_ONION 1
#define OC_O_TOMATO 2
#define OC_O_DISH 3
#define OC_O_SOUP 0x80

/* per-env flag byte written by oc_step / oc_rollout_random */
#define OC_F_DONE 0x01       /* timestep >= horizon after this step (env.py:321-325) */
#define OC_F_BAD_ACTION 0x02 /* action index >= 6: state left untouched (mdp.py:1394-1398 raises) */
#define OC_F_RESET 0x04      /* env was auto-reset to its start state after this step */

/* option bits for oc_step / oc_rollout_random */
#define OC_OPT_AUTO_RESET 0x1u /* reset a done env to its layout's start state inside the kernel */

/* 0x2u: retired (was OC_OPT_LANE_PER_ENV; one lane per env is the default and only automatic choice); ignored when set */
#define OC_OPT_LANE_PAIR 0x4u    /* oc_rollout_random: the lane-pair-per-env kernel where the table allows it
                                   (2-player layouts, <= 2 pots); never chosen automatically */
#define OC_OPT_PREDICATE_INTERACT 0x8u /* oc_rollout_random: one-lane-per-env kernel with the predicate-network
                                         interact instead of the table-driven one (kept for cross-checking) */

/* 0x10u: reserved (ABI <= 3: OC_OPT_ROLLOUT_V3, the round-1 rollout kernel, retired in ABI 4) */
#define OC_OPT_ONE_KERNEL 0x20u /* oc_rollout_encode / oc_step_encode: take the single-kernel path (k_rollout_encode)
                                  whenever the table allows it; by default it runs only for batches that give every CU
                                  a workgroup (it keeps 256 envs per CU on chip; smaller batches are faster through the
                                  one-step kernels, whose observation kernel spreads over all CUs) */
#define OC_OPT_FLAGS_TILED8 0x40u /* oc_rollout_random: d_flags is tiled by 8 steps, [n_steps / 8][n_envs][8] — byte
                                     (k / 8, e, k % 8) holds the OC_F_* bits of env e after step k of the call — so that a
                                     wavefront writes the flags of 8 steps as 512 contiguous bytes instead of 64 bytes in each
                                     of 8 rows (worth ~6 % of the rollout rate and its run-to-run spread on MI355X).  Needs
                                     t0 and n_steps to be multiples of 8, d_rewards and an 8-byte aligned d_flags, no event
                                     sink, and a batch one of these kernels serves: the pipelined joint-table kernel (one
                                     two-player, one-pot layout with at most 6 free cells and no shared faced cells —
                                     cramped_room —, at most ~98 000 envs) or the per-env-terrain kernels of mixed two-player
                                     tables (up to 32 layouts: at most ~98 000 envs; one-pot tables of more layouts: any
                                     batch size) — BASELINE configs[1], [3], [4] —, or the mover / interact kernel described
                                     under OC_OPT_ONE_WAVEFRONT (single two-player layouts included; batches of whole 256-env
                                     workgroups up to 524 288 envs); OC_EINVAL otherwise, so a caller can try it once and
                                     fall back */
#define OC_OPT_ONE_WAVEFRONT 0x80u /* oc_rollout_random: keep every env-step in ONE wavefront.  Without this bit, batches of
                                     two-player layouts with at most two pots and 64 cells (new dynamics, one set of shaping
                                     rewards, both output arrays, no event sink) that consist of whole 256-env workgroups —
                                     one per CU at a time: batches above 65 536 envs (MI355X) run in up to 8 rounds, tables
                                     read through L2 from the third round on — and are launched over whole 8-step blocks (t0
                                     and n_steps multiples of 8) run with the step split between two wavefronts per 64 envs: a
                                     mover (actions, resolve_movement, horizon, flag bytes) running up to 16 steps ahead of
                                     an interact wavefront (resolve_interacts, env effects, rewards) through a ring in LDS.
                                     Same results bit for bit; the bit exists so that callers (and the parity tests) can
                                     cross-check the two formulations, as with OC_OPT_LANE_PAIR / OC_OPT_PREDICATE_INTERACT */

/* OcBatch.batch_flags */
#define OC_BATCH_TWO_PLAYERS 0x1u /* every layout of the table has exactly 2 players */
#define OC_BATCH_NEW_DYNAMICS 0x2u /* no layout of the table uses old_dynamics (mdp.py:1696-1701) */
#define OC_BATCH_NO_SHARED_FACES 0x8u /* in no layout of the table does a non-floor cell touch two floor cells: two players can
                                         never face the same cell (cramped_room), so the interact order needs no replay */
#define OC_BATCH_UNIFORM_SHAPING 0x4u /* every layout of the table has the same rew_shaping_params and old_dynamics flag:
                                         the interact table then carries the reward floats for the whole batch */

/* obs dtypes of oc_encode_lossless */
#define OC_OBS_U8 0
#define OC_OBS_F32 1

/* error codes */
#define OC_OK 0
#define OC_EINVAL (-1)
#define OC_ELAUNCH (-2)

/*
 * One compiled layout: terrain + the recipe/reward configuration of one OvercookedGridworld
 * (mdp.py:1090-1148) flattened to look-up tables.  256 bytes, 16-byte aligned.  The table is
 * built on the host by overcooked_ai_amd.layouts.compile_layout and uploaded once.
 *   cook_time[n_onion + 4*n_tomato]      = Recipe.time  (mdp.py:163-188)
 *   delivery_value[n_onion + 4*n_tomato] = get_recipe_value, non-discounted branch (mdp.py:1595-1602):
 *        0 if the recipe is not in all_orders, order_bonus*value if it is a bonus order, else value.
 *   terrain[c] = terrain code | (pot slot << 3) for cell c = y*W + x  (terrain_mtx[y][x], mdp.py:1783)
 */
typedef struct OcLayout {
    uint8_t width, height, n_cells, n_pots;
    uint8_t n_players, old_dynamics, n_obj_planes, reserved0;
    uint8_t start_pos[2];
    uint8_t start_or[2];
    uint8_t reserved1[4];
    uint8_t pot_cell[OC_MAX_POTS];
    uint8_t potting_class[8];   /* [n_onion + 3*n_tomato of the soup BEFORE the potting], low nibble: an onion is
                                   added, high nibble: a tomato; bit 0 optimal, 1 viable, 2 catastrophic, 3 useless
                                   (is_potting_*, mdp.py:2256-2308) */
    float rew_placement_in_pot; /* PLACEMENT_IN_POT_REW, mdp.py:1019 */
    float rew_dish_pickup;      /* DISH_PICKUP_REWARD */
    float rew_soup_pickup;      /* SOUP_PICKUP_REWARD */
    float reserved3;
    uint8_t cook_time[16];
    float delivery_value[16];
    uint8_t terrain[OC_MAX_CELLS];
} OcLayout;

/*
 * A batch of envs: which layouts they use and how many there are.  Host-side POD; the two d_ members
 * are device pointers.  All layouts of one table share width/height (pad smaller grids with counters,
 * as the reference's LayoutGenerator.embed_grid does, layout_generator.py:309-329).
 */
typedef struct OcBatch {
    const OcLayout* d_layouts;   /* [n_layouts] compiled layouts in HBM */
    const uint16_t* d_layout_id; /* [n_envs] layout index per env, or NULL when n_layouts == 1.  WRITTEN by the step kernels
                                    (and oc_regen_layouts) when an OcStartSpec with regen_count > 0 moves restarted envs
                                    to other layouts: keep it in writable device memory then */
    int64_t n_envs;
    int32_t n_layouts;
    int32_t width, height;       /* grid shape shared by every layout of the table */
    int32_t max_pots;            /* max n_pots over the table (1..8), or 0 if unknown: selects how many pot slots
                                    the step kernels keep in registers */
    uint32_t batch_flags;        /* OC_BATCH_* hints about the table */
    uint32_t max_free_cells;     /* max number of free (floor) cells over the table, or 0 if unknown: single two-player
                                    layouts with at most 7 free cells (cramped_room) step through a joint move table */
} OcBatch;

/*
 * max_pots, batch_flags and max_free_cells are HINTS that select kernel variants, and hard preconditions when set:
 * a max_pots below a layout's n_pots drops that layout's extra pots, OC_BATCH_TWO_PLAYERS on a one-player layout or a
 * max_free_cells below the real count index past the tables.  0 / unset is always safe.  oc_batch_hints derives them
 * from a HOST copy of the layout table (call it before uploading the table; it leaves the other fields alone).
 */
int oc_batch_hints(const OcLayout* h_layouts, int n_layouts, OcBatch* batch);

/*
 * How a finished env is restarted by OC_OPT_AUTO_RESET (and by oc_multi_agent_step): NULL = the standard start state
 * (get_standard_start_state, mdp.py:1297-1305); otherwise the start_state_fn of
 * OvercookedGridworld.get_random_start_state_fn(random_start_pos, rnd_obj_prob_thresh) (mdp.py:1307-1369) as
 * OvercookedEnv.reset uses it (env.py:288-319), drawn inside the step kernel with the stream documented at
 * oc_reset_random for global env g = env_offset + e and
 *      epoch of a restart at step k of the call = epoch + k      (k = 0 for single-step entry points),
 * so a caller that passes epoch = 1 + (steps executed so far) never reuses the draws of its initial
 * oc_reset_random(epoch 0).  Supported by the table-driven kernels (oc_step and oc_step_many without
 * OC_OPT_PREDICATE_INTERACT — with or without event logging —, oc_rollout_random without OC_OPT_LANE_PAIR /
 * PREDICATE_INTERACT, oc_step_encode, oc_rollout_encode, oc_multi_agent_step); the others return OC_EINVAL.
 */
typedef struct OcStartSpec {
    uint64_t seed;
    int64_t env_offset;          /* global index of local env 0 (oc_rollout_random / oc_rollout_encode: must equal their
                                    env_offset argument) */
    uint32_t epoch;
    int32_t random_start_pos;    /* 0 / 1 */
    double rnd_obj_prob_thresh;  /* 0 .. 1 */
    /* Per-episode layout re-draw = OvercookedEnv.reset(regen_mdp=True) over an mdp_generator_fn that yields a different
     * layout every time (env.py:288-302; LayoutGenerator, layout_generator.py:110-160 — BASELINE configs[4]): with
     * regen_count > 0 an env that restarts first moves to layout
     *      regen_first + mulhi32(word 0 of block 15 of the reset stream of oc_reset_random, regen_count)
     * (same epoch as its start-state draws), the new id is stored in OcBatch.d_layout_id[e], and the start state (standard,
     * or drawn per the two fields above) is that of the NEW layout.  regen_first + regen_count <= n_layouts; 0 = off. */
    uint32_t regen_first;
    uint32_t regen_count;
} OcStartSpec;

/*
 * Where the event_infos of a call go (EVENT_TYPES, mdp.py:1027-1058; log_object_* / is_*_useful / is_potting_*,
 * mdp.py:2121-2308); NULL = no event logging.  Bit 2*k + p of a mask is event_infos[EVENT_TYPES[k]][p].
 *   d_events       [n_steps][n_envs] u64: the mask of every step, or NULL
 *   d_counts       [n_envs][25] u32: how often each event happened in the env's RUNNING episode — player 0 in bits
 *                  0..15, player 1 in bits 16..31 (the lengths of game_stats[event][player], env.py:382-401; what
 *                  RLlib reports per agent, human_aware_rl/rllib/rllib.py:453-483).  Accumulated across calls by the
 *                  step kernels; zeroed when the env is auto-reset.  Or NULL
 *   d_counts_done  [n_envs][25] u32: the counts of the last FINISHED episode (ep_game_stats), written at the step that
 *                  ends it, or NULL
 */
typedef struct OcEventSink {
    uint64_t* d_events;
    uint32_t* d_counts;
    uint32_t* d_counts_done;
} OcEventSink;

int oc_abi_version(void);
size_t oc_layout_size(void); /* == sizeof(OcLayout) == 256 */
const char* oc_last_error(void);
/* number of 16-byte planes per env for a width x height grid: 1 + ceil(width*height/16) */
int oc_state_planes(int width, int height);

/*
 * oc_step — one joint transition for n_envs independent envs.
 * Replaces OvercookedGridworld.get_state_transition (mdp.py:1375-1430) = resolve_interacts (1432)
 * -> resolve_movement (1644) -> step_environment_effects (1691), plus OvercookedEnv.step's
 * done/bookkeeping (env.py:244-274, 321-325, 382-392).
 *   d_state_in/out [oc_state_planes * n_envs] 16-byte words; may alias (in-place step)
 *   d_actions      [n_envs][2] action indices 0..5 (player 0, player 1)
 *   d_rewards      [n_envs][4] float: sparse_reward_by_agent[0..1], shaped_reward_by_agent[0..1]
 *   d_flags        [n_envs] OC_F_* bits
 *   d_ep_returns   [n_envs][4] float running sums of d_rewards over the episode
 *                  (game_stats cumulative_*_rewards_by_agent, env.py:387-392), or NULL
 *   d_events       [n_envs] u64 event_infos of this step (EVENT_TYPES, mdp.py:1027-1058): bit 2*k + p is
 *                  event_infos[EVENT_TYPES[k]][p]; or NULL (shorthand for an OcEventSink with only d_events)
 *   events         per-episode event counters / masks (OcEventSink), or NULL
 *   horizon        done when timestep >= horizon (1..65535; the stored timestep is a u16 that saturates at 65535 — the
 *                  Python OvercookedEnv keeps the reference's default horizon of 1e10 but raises at that limit)
 * In place (d_state_out == d_state_in) without event logging, on grids of at most 64 cells, the step runs on the wire format
 * itself (k_step1: header + changed object bytes written back, <= 4.3 us per batched step of 65 536 envs, launch-bound); other
 * forms unpack the env (k_step3, 5.7-7.0 us).  Same results.
 */
int oc_step(const OcBatch* batch, const void* d_state_in, void* d_state_out, const uint8_t* d_actions,
            float* d_rewards, uint8_t* d_flags, float* d_ep_returns, uint64_t* d_events, int horizon,
            uint32_t options, const OcStartSpec* start, const OcEventSink* events, void* stream);

/*
 * oc_step_many — n_steps consecutive oc_step transitions (in place) in ONE launch: step k consumes
 * d_actions[k][n_envs][2] and writes d_rewards[k][n_envs][4], d_flags[k][n_envs]; the envs stay on chip between the
 * steps (replaying a fixed joint plan or a pre-sampled action tensor: OvercookedEnv.execute_plan, env.py:334-345;
 * AgentEvaluator._check_trajectories_dynamics, benchmarking.py:366).  Results are those of n_steps oc_step calls.
 */
int oc_step_many(const OcBatch* batch, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                 float* d_ep_returns, int n_steps, int horizon, uint32_t options, const OcStartSpec* start,
                 const OcEventSink* events, void* stream);

/*
 * oc_rollout_random — n_steps transitions per launch under the uniform random policy
 * (the reference's RandomAgent(all_actions=True) pair, agents/agent.py:223), actions drawn
 * in-kernel.  One Philox4x32-10 block feeds 8 consecutive steps: with b = t >> 3, s = t & 7,
 *      r[0..3] = philox4x32_10(counter = {b_lo, g_lo, g_hi, b_hi}, key = {seed_lo, seed_hi}),
 *      x = r[s >> 1] * (s & 1 ? 36 : 1)  (mod 2^32),
 *      action of player 0 = mulhi32(x, 6),  action of player 1 = mulhi32(x * 6 mod 2^32, 6)
 * for global env g at global step t (base-6 digits of the 32-bit word; bias < 6^4 / 2^32).
 * Same transition function as oc_step; state stays on chip between the fused steps.  A launch costs ~16 us outside its
 * step loop (table staging, state load / store, dispatch): 12 % of a 400-step launch of 65 536 envs, 1.5 % of a 4 000-step
 * one (207 vs 244 G env-steps/s on MI355X) — prefer few long launches.
 * Launch shape and speed: the mover / interact kernel works in whole 8-step blocks of whole 256-env workgroups; a long call off that
 * step grid (t0 or n_steps not a multiple of 8) is split inside the call — head and tail steps through the one-wavefront kernels —, a
 * ragged batch takes the one-wavefront kernels altogether (oc_rollout_plan says which instance a call takes).
 *   d_rewards  [n_steps][n_envs][4] or NULL;  d_flags [n_steps][n_envs] or NULL ([n_steps / 8][n_envs][8] with OC_OPT_FLAGS_TILED8)
 *   env_offset global index of local env 0 (multi-GPU shards draw disjoint streams)
 *   t0         global step index of the first fused step
 */
int oc_rollout_random(const OcBatch* batch, void* d_state, float* d_rewards, uint8_t* d_flags,
                      float* d_ep_returns, int horizon, uint32_t options, uint64_t seed,
                      int64_t env_offset, int64_t t0, int n_steps, const OcStartSpec* start,
                      const OcEventSink* events, void* stream);

/*
 * oc_encode_lossless — the 26-layer observation of both players.
 * Replaces OvercookedGridworld.lossless_state_encoding (mdp.py:2385-2561) as called through
 * OvercookedEnv.lossless_state_encoding_mdp (env.py:276-280).
 *   d_obs  [n_envs][2][W][H][26] of u8 or f32 (index [x][y][layer], mdp.py:2415-2418, 2550); 16-byte aligned
 */
int oc_encode_lossless(const OcBatch* batch, const void* d_state, void* d_obs, int obs_dtype, int horizon,
                       void* stream);

/*
 * oc_step_encode — oc_step (in place, caller's actions, auto-reset per `options`, drawn start states per `start`)
 * followed by oc_encode_lossless of the resulting states, i.e. the step of a training / evaluation loop that feeds the
 * lossless observation to a policy: OvercookedEnv.step (env.py:244) + lossless_state_encoding_mdp of the state the next
 * step starts from (env.py:276; human_aware_rl/rllib/rllib.py:257-260).  One C call enqueues the two kernels back to
 * back on `stream`; with OC_OPT_ONE_KERNEL (and no `start`) it is oc_rollout_encode with n_steps = 1 — a wash for a
 * single step (36.4 vs 37.2 us on 65 536 asymmetric_advantages envs, 25.1 vs 24.4 us on cramped_room), the gain of the
 * single kernel comes with several steps per launch.  Arguments as in oc_step / oc_encode_lossless.
 */
int oc_step_encode(const OcBatch* batch, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                   float* d_ep_returns, void* d_obs, int obs_dtype, int horizon, uint32_t options,
                   const OcStartSpec* start, void* stream);

/*
 * oc_rollout_encode — n_steps transitions AND the lossless observation after each of them, in one call: BASELINE
 * configs[2] (the rollout of configs[1] "plus oc_encode_lossless every step"), i.e. a trajectory of
 * (reward, flag, observation) per step as a rollout collector (OvercookedEnv.run_agents / get_rollouts, env.py:425-580,
 * over step 244 + lossless_state_encoding_mdp 276) gathers it.
 *   d_actions  NULL: the uniform random policy, the Philox stream of oc_rollout_random (seed, env_offset, t0);
 *              else [n_steps][n_envs][2] action indices as for oc_step_many (illegal: flagged, env untouched)
 *   d_rewards  [n_steps][n_envs][4] / d_flags [n_steps][n_envs] (may be NULL with the random policy)
 *   d_obs      observation of step k at (char*)d_obs + k * obs_step_stride: [n_envs][2][W][H][26] of obs_dtype, the state
 *              the NEXT step starts from (after an auto-reset: the start state), exactly what oc_step_encode emits;
 *              obs_step_stride in bytes, a multiple of 16; 0 = every step overwrites the same observation
 *   options    OC_OPT_AUTO_RESET, OC_OPT_ONE_KERNEL;  start: as for oc_rollout_random (NULL = standard start states; a
 *              restart at step k draws from epoch start->epoch + k)
 * One layout, at most two pots, at least two steps and a batch that fills the GPU run as ONE kernel
 * (k_rollout_encode: the env stays on chip for all steps, every wavefront encodes its own 64 envs through a private
 * LDS image, no workgroup barrier in the step loop): 30 us per step on 65 536 asymmetric_advantages envs, the rate at
 * which the observation bytes alone reach HBM (two one-step kernels: 37 us).  Any other case runs the one-step kernels
 * step by step with identical results.
 */
int oc_rollout_encode(const OcBatch* batch, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                      float* d_ep_returns, void* d_obs, int obs_dtype, int64_t obs_step_stride, int horizon,
                      uint32_t options, uint64_t seed, int64_t env_offset, int64_t t0, int n_steps,
                      const OcStartSpec* start, void* stream);

/*
 * oc_featurize — the hand-crafted feature vector of both players.
 * Replaces OvercookedGridworld.featurize_state (mdp.py:2579-2898) as called through
 * OvercookedEnv.featurize_state_mdp (env.py:282-286); needs 2-player layouts.
 *   d_plan_blob / d_plan_off  per-layout motion-cost tables built on the host by overcooked_ai_amd.planner
 *       (the MotionPlanner distances of planning/planners.py:391-423): at byte d_plan_off[layout] of the blob,
 *       floor_index[128] (cell -> index among the free cells, row-major) followed by
 *       cost[(floor_index[cell] * 4 + orientation) * row_stride + feature_cell] = fewest actions to stand next to the
 *       feature facing it, 255 = unreachable or not a motion goal (counters outside MotionPlanner.counter_goals);
 *       row_stride = n_cells rounded up to a multiple of 16 (rows are fetched as 16-byte words), padding = 255.
 *       ABI 4: d_plan_off holds 2 * n_layouts entries; at byte d_plan_off[n_layouts + layout] starts the layout's WALK
 *       SECTION, {u32 record_stride, 12 pad bytes} + one record per (free cell, orientation) state: what a walk over the
 *       whole grid would find as far as it depends on the terrain alone — u32[4] arg-min keys (cost << 9 | cell,
 *       0xFFFFFFFF = none) of the closest onion / tomato / dish dispenser and serving cell, u32[4] the four closest pots in
 *       ascending key order, u8 n + u8[n] the goal counters in ascending (cost, cell) order (planner.walk_records)
 *   d_features  [n_envs][2][2 * (num_pots * 10 + 26) + 4] float32, 16-byte aligned; row i = features for player i
 *   num_pots    0..4 (the reference's default is 2 -> 96 features)
 * Ties between equally cheap counter objects are broken by cell order (row-major); the reference breaks them by the
 * insertion order of its objects dict, which the packed state does not carry.
 */
int oc_featurize(const OcBatch* batch, const uint8_t* d_plan_blob, const uint32_t* d_plan_off, const void* d_state,
                 float* d_features, int num_pots, void* stream);

/*
 * oc_shape_rewards — the per-agent training reward of the RLlib environment.
 * Replaces the reward arithmetic of OvercookedMultiAgent.step (human_aware_rl/rllib/rllib.py:306-329):
 *      out[e][i] = (sparse0 + sparse1) + reward_shaping_factor * dense_i
 * with dense_i = phi_next[e] - phi_cur[e] for both agents when d_phi_next != NULL (use_phi), else the shaped
 * reward of agent i from d_rewards.  float64, like the reference's Python floats.
 *   d_rewards [n_envs][4], d_flags [n_envs]   as written by oc_step
 *   d_phi_next [n_envs] phi(s') (oc_potential after the step, before any reset), or NULL
 *   d_phi_cur  [n_envs] in: phi(s); out: the potential of the state the next step starts from — phi(s'), or
 *              d_phi_start[layout] where the episode is done (the caller resets those envs with d_done as the mask)
 *   d_phi_start [n_layouts] potential of each layout's standard start state
 *   d_out  [n_envs][2] float64, 16-byte aligned;  d_done [n_envs] 1 where OC_F_DONE is set, or NULL
 */
int oc_shape_rewards(const OcBatch* batch, const float* d_rewards, const uint8_t* d_flags, const double* d_phi_next,
                     double* d_phi_cur, const double* d_phi_start, double reward_shaping_factor, double* d_out,
                     uint8_t* d_done, void* stream);

/*
 * oc_multi_agent_step — one step of the RLlib training environment, enqueued by a single call.
 * Replaces OvercookedMultiAgent.step (human_aware_rl/rllib/rllib.py:293-342) for a batch: oc_step (no auto-reset) ->
 * oc_potential on s' (when d_phi_tables != NULL: use_phi) -> oc_shape_rewards -> copy of the episode returns ->
 * oc_reset of the finished envs (mask = d_done) -> oc_encode_lossless of the states the next step starts from
 * (when d_obs != NULL).  Arguments as in those entry points; d_done is required.  Two-player tables with at most two
 * pots run everything before the encoding as one kernel (k_train_step) with identical results.  With `start`, finished
 * envs restart from drawn start states and d_phi_cur receives the potential of those.
 * ABI 5, round 5: with d_obs, ONE layout on a grid of at most 64 cells, no event sink and a batch that gives at least half
 * of the CUs a workgroup (>= 32 768 envs on MI355X) the whole call is ONE kernel (k_train_step_obs, csrc/train_obs.hpp: eight
 * wavefronts per 256 envs — four step and restart, four compute phi and the shaped rewards, all eight encode and stream the
 * observations): 65 536 cramped_room envs with u8 observations 26.2 -> 20.3 us per call, asymmetric_advantages 37.6 -> 30.7 us
 * (profiles/r05_train_step_obs.txt); same results bit for bit.
 */
int oc_multi_agent_step(const OcBatch* batch, void* d_state, const uint8_t* d_actions, float* d_rewards,
                        uint8_t* d_flags, float* d_ep_returns, float* d_ep_returns_out, const uint8_t* d_plan_blob,
                        const uint32_t* d_plan_off, const uint8_t* d_phi_tables, double* d_phi_next,
                        double* d_phi_cur, const double* d_phi_start, double reward_shaping_factor, double* d_shaped,
                        uint8_t* d_done, void* d_obs, int obs_dtype, int horizon, const OcStartSpec* start,
                        const OcEventSink* events, void* stream);

/*
 * oc_reset_random — randomized start states drawn on the GPU.
 * Replaces the start_state_fn of OvercookedGridworld.get_random_start_state_fn(random_start_pos,
 * rnd_obj_prob_thresh) (mdp.py:1307-1369) as used by OvercookedEnv.reset (env.py:288-319) for training-time
 * diversity.  Same distribution as the reference: joint positions uniform over ordered tuples of distinct free cells
 * (orientation NORTH), each pot filled with probability thresh (n = 1..3 onions, then 0..3-n tomatoes; cooking from
 * tick 0 with probability thresh, else idle), each player holding an object with probability thresh (dish 0.2,
 * onion 0.6, finished soup 0.2 with the same ingredient draw).  The draws come from this library's counter-based
 * stream, not from numpy's global generator (the numpy-exact version stays on the host:
 * overcooked_ai_amd.mdp.OvercookedGridworld.get_random_start_state_fn):
 *      block(b) = philox4x32_10(counter = {epoch, g_lo, g_hi, b}, key = {seed_lo, seed_hi ^ 0x52535421})
 *      b = 0: word 0 -> joint position index mulhi(word, n_joint) into the row-major product of free cells
 *      b = 1 + i (player i): {u, kind, n, m}: holds iff u < T; kind < 858993459 dish, < 3435973836 onion, else soup;
 *                            n_onion = 1 + mulhi(n, 3), n_tomato = mulhi(m, 4 - n_onion)
 *      b = 3 + k (pot k):    {u, n, m, q}: filled iff u < T; cooking (tick 0) iff q < T
 *      T = floor(rnd_obj_prob_thresh * 2^32)
 * for global env g = env_offset + e.  d_mask as in oc_reset.
 */
int oc_reset_random(const OcBatch* batch, void* d_state, const uint8_t* d_mask, float* d_ep_returns, uint64_t seed,
                    int64_t env_offset, uint32_t epoch, int random_start_pos, double rnd_obj_prob_thresh,
                    void* stream);

/*
 * oc_regen_layouts — new layout ids for the envs an explicit reset is about to restart (regen_mdp=True semantics for
 * oc_reset / oc_reset_random: call this first, then the reset with the same mask).  Env e is selected when d_mask is NULL
 * or (d_mask[e] & mask_bits) != 0 — mask_bits = OC_F_RESET selects from a flags array the envs a step has just restarted;
 * its new id is the draw documented at OcStartSpec for (start->seed, start->env_offset + e, start->epoch).
 *   d_layout_id  [n_envs], writable; start->regen_count > 0 required
 */
int oc_regen_layouts(const OcBatch* batch, uint16_t* d_layout_id, const uint8_t* d_mask, uint8_t mask_bits,
                     const OcStartSpec* start, void* stream);

/*
 * oc_potential — phi(s), the potential used for potential-based reward shaping.
 * Replaces OvercookedGridworld.potential_function(state, mp, gamma) (mdp.py:2920-3238) as called by
 * get_state_transition(display_phi=True) (mdp.py:1421-1429) for `phi_s` / `phi_s_prime`, which the RLlib
 * environment turns into the dense reward gamma * phi_s_prime - phi_s (human_aware_rl/rllib/rllib.py:314-319).
 *   d_plan_blob / d_plan_off  the motion-cost tables of oc_featurize (any counter_goals: only pots and serving
 *       cells are looked up)
 *   d_phi_tables [n_layouts][oc_phi_table_size()] per-layout records for ONE discount factor, built on the host by
 *       overcooked_ai_amd.potential.pack_phi_tables (layout documented there): steady-state value, the best
 *       completion of every ingredient multiset (_get_optimal_possible_recipe, mdp.py:1976-2016), gamma ** k;
 *       8-byte aligned
 *   d_phi  [n_envs] float64
 * Arithmetic is float64 in the reference's operand order without contraction: results are bit-identical to the
 * reference under CPython 3.8-3.12 (whose set iteration order decides ties between partially full pots,
 * mdp.py:1882-1890).
 */
int oc_potential(const OcBatch* batch, const uint8_t* d_plan_blob, const uint32_t* d_plan_off,
                 const uint8_t* d_phi_tables, const void* d_state, double* d_phi, void* stream);
int oc_phi_table_size(void);

/*
 * oc_reset — write the standard start state (OvercookedGridworld.get_standard_start_state,
 * mdp.py:1297-1305: players at start positions facing NORTH, no objects, timestep 0) into every
 * env whose d_mask byte is non-zero (all envs when d_mask is NULL).  Replaces OvercookedEnv.reset
 * (env.py:288-319) for the default start_state_fn.  d_ep_returns (nullable) is zeroed for reset envs.
 */
int oc_reset(const OcBatch* batch, void* d_state, const uint8_t* d_mask, float* d_ep_returns, void* stream);

/*
 * oc_mailbox_* — ONE env stepped per call without a kernel launch per call: what the reference's
 * OvercookedEnv.step (overcooked_env.py:244) -> OvercookedGridworld.get_state_transition (overcooked_mdp.py:1375) does for
 * an agent loop.  oc_mailbox_open starts a resident one-wavefront kernel that serves transitions of the batch's single
 * layout from a 4 KiB mailbox in pinned, GPU-mapped host memory; the caller writes the packed state (oc_state_planes()
 * planes of 16 bytes, as [plane][16]) at OC_MB_STATE_IN and the two action indices at OC_MB_ACTIONS of oc_mailbox_buffer(),
 * calls oc_mailbox_step (which posts the request and spins until the kernel has answered: ~7 us instead of the ~16 us of a
 * launch + stream wait) and reads the next state at OC_MB_STATE_OUT, float rewards[4] = (sparse0, sparse1, shaped0, shaped1) at
 * OC_MB_REWARDS, the OC_F_* flags (u32; DONE when the new timestep >= horizon, BAD_ACTION leaves the state as it was) at
 * OC_MB_FLAGS and the event_infos mask (u64, bit 2*k + p) at OC_MB_EVENTS.  Same transition, bit for bit, as oc_step.
 * The kernel leaves on its own after ~2 ms without a request (and after ~2 s in any case) and is relaunched by the next
 * oc_mailbox_step, so a device-wide synchronisation elsewhere waits at most that long.  One layout (batch.n_layouts == 1),
 * grids of at most 64 cells; not thread-safe per mailbox.  Returns OC_EINVAL for other batches (use oc_step).
 */
typedef struct OcMailbox OcMailbox;
#define OC_MB_STATE_IN 256
#define OC_MB_ACTIONS 336
#define OC_MB_STATE_OUT 512
#define OC_MB_REWARDS 592
#define OC_MB_FLAGS 608
#define OC_MB_EVENTS 616
int oc_mailbox_open(const OcBatch* batch, int horizon, OcMailbox** mailbox);
void* oc_mailbox_buffer(OcMailbox* mailbox);
int oc_mailbox_step(OcMailbox* mailbox);
int oc_mailbox_close(OcMailbox* mailbox);

/*
 * The resident batched step (ABI 6) — OvercookedEnv.step (overcooked_env.py:244-274) for the whole batch WITHOUT a launch per step.
 * oc_step costs a dependent kernel boundary plus its own load -> transition -> store chain per call (4.5 us for 65 536 envs); a
 * caller that lives on the GPU (a persistent policy kernel, the last kernel of a forward pass) can instead talk to a resident
 * kernel that keeps the envs in registers / LDS between steps, through per-env mailboxes in device memory:
 *   request   uint64 [n_envs]     low word  a0 | a1 << 8 (| OC_SV_STOP) [| caller's XCD << 20 | 1 << 24: lets the server look
 *                                 through its L2 between device-scope looks when both ends share an XCD], high word = tag;
 *                                 ONE aligned 8-byte store per env,
 *                                 written through to device scope (gfx950: `global_store_dwordx2 ... sc1`)
 *   response  uint32 [n_envs][8]  {sparse0, sparse1, shaped0 (float bits), tag} {shaped1, flags (OC_F_*), info, tag}
 *                                 info = timestep (bits 0..15) | XCD the server's workgroup runs on (20..23) | 1 << 24;
 *                                 two aligned 16-byte stores by the server; a granule that shows the tag is complete
 * tag = 1, 2, 3, ... = the number of the step since the server was opened (every env is sent every step; a wavefront steps when
 * its 64 envs all show the next tag).  Readers poll with device-scope loads (`sc1`): plain loads may be served from the reader's
 * own L2.  The transition, the bookkeeping, OC_OPT_AUTO_RESET and the drawn starts of `start` (restart at step k: epoch
 * start->epoch + k - 1, as k consecutive oc_step calls) are oc_step's, bit for bit (tests/test_gpu_step_server.py).
 * While the server is resident the states live on chip: d_state / d_ep_returns are current again after oc_step_server_sync (or
 * close, or once the kernel has left by itself).  The kernel leaves after idle_ms without a request (0: 20 ms) and after life_s in
 * any case (0: 600 s), writing the states back; oc_step_server_resume / _play relaunch it.  Device-wide synchronisation
 * (hipDeviceSynchronize, torch.cuda.synchronize()) waits for it to leave — synchronise streams or events instead.
 * Not served: OcEventSink, OC_OPT_PREDICATE_INTERACT; the batch's workgroups (n_envs / 256) must fit the GPU at once.
 *   oc_step_server_play  the caller's side as a kernel on `stream` (k_step_client: lane = env; per step post the request, poll the
 *                        response, leave rewards [n_steps][n_envs][4] / flags [n_steps][n_envs]) for actions uint8
 *                        [n_steps][n_envs][2] — n_steps = 1: a drop-in oc_step; n_steps > 1: the parity tests' replay of
 *                        oc_step_many's inputs and bench.py's round-trip measurement.  Host-synchronous; elapsed_ms (or NULL)
 *                        receives the client kernel's duration (HIP events on `stream`).
 *   oc_step_server_steps steps served so far (the host's count; exact after _play / _sync)
 * One caller at a time: the entry points of a server are not thread-safe, and two clients must not play on it concurrently.
 * The resident kernel reads d_state / d_ep_returns on a stream of its own whenever it is (re)launched — at _open, and at a _play /
 * _resume that finds it gone: work of the caller's on those arrays must be complete by then (_play waits for `stream` before a
 * relaunch; before _open and _resume the caller synchronises its stream itself).
 */
#define OC_SV_STOP 0x10000u
typedef struct OcStepServer OcStepServer;
int oc_step_server_open(const OcBatch* batch, void* d_state, float* d_ep_returns, int horizon, uint32_t options,
                        const OcStartSpec* start, double idle_ms, double life_s, OcStepServer** server);
void* oc_step_server_requests(OcStepServer* server);  /* device pointer, uint64 [n_envs] */
void* oc_step_server_responses(OcStepServer* server); /* device pointer, uint32 [n_envs][8] */
int oc_step_server_resume(OcStepServer* server);
int oc_step_server_play(OcStepServer* server, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags, int n_steps,
                        void* stream, float* elapsed_ms);
int oc_step_server_sync(OcStepServer* server);
int64_t oc_step_server_steps(OcStepServer* server);
int oc_step_server_close(OcStepServer* server);

/*
 * Measurement aid (round 4): nothing but oc_rollout_random's OUTPUT STORES — one lane per env, per step one reward quad
 * (16 bytes, zeros) at d_rewards[k][e] and one flag byte (0) at d_flags[k][e], same workgroup shape and row addressing as the
 * rollout kernels, no state, no game.  Timing it says what the [step][env] output format of oc_rollout_random admits on the
 * device at this batch size (MI355X, 65 536 envs: 370-375 G env-steps/s = 0.79-0.80 of the HBM peak on one box, and
 * sensitive to the loop around the stores: 352 G for tools/store_rate.hip's loop on the same box), i.e. the ceiling bench.py
 * reports next to the roofline.  d_flags may be NULL (quads only).  Leaves the arrays zeroed.
 * options (ABI 5): 0, or OC_OPT_FLAGS_TILED8 — the flags array in the tiled layout ([n_steps / 8][n_envs][8]: per 8-step block
 * one 8-byte store per lane into the block's tile row, as the kernels serving that layout write it; n_steps a multiple of 8,
 * d_flags 8-byte aligned and not NULL), so that the ceiling is reported in the layout the rollout was timed in.
 */
int oc_output_stores_only(int64_t n_envs, int n_steps, float* d_rewards, uint8_t* d_flags, uint32_t options, void* stream);

/*
 * oc_rollout_plan (ABI 6) — which kernel instance oc_rollout_random would launch for this batch and launch shape, as text
 * (e.g. "k_rollout5<LAY_LDS=true, FT8=true, OLD=false, BIG=false, EV=false> mover + interact wavefronts, 1 round(s), 130864 B LDS").
 * The answer comes from oc_rollout_random's own dispatch, walked with stand-in pointers: every argument check applies, every branch
 * is the one a real call takes, nothing is launched and no device memory is touched — so it also runs on a host without a GPU
 * (the device's SIMD count then defaults to MI355X's 1 024).  docs/DISPATCH.md is generated from it (tools/gen_dispatch_table.py)
 * and tests/test_dispatch_table.py keeps that file equal to what the library answers.
 *   with_outputs  1: d_rewards and d_flags are given; 0: both NULL
 *   event_sink    0 none, 1 per-episode counters (OcEventSink.d_counts), 2 per-step masks as well (d_events)
 *   start         NULL or the start-state description the call would carry
 *   out, out_size caller's text buffer (>= 256 bytes holds every answer)
 */
int oc_rollout_plan(const OcBatch* batch, int horizon, uint32_t options, int64_t t0, int n_steps, int with_outputs,
                    int event_sink, const OcStartSpec* start, char* out, size_t out_size);

#ifdef __cplusplus
}
#endif
#endif /* OC_AMD_H */
