// step_duo5.hpp — get_state_transition, fifth formulation: k_rollout5, the per-env-terrain mover / interact kernel (round 6)
// Part of liboc_amd.so: included by rollout4.hip (OC_R4_PART 1) inside its anonymous namespace after step_lut4.hpp, whose helpers
// (LDS accessors by absolute address, the Philox block, the flag-byte store) it shares.
#pragma once

// ==========================================================================================
// Why a fifth formulation.  Round 6 measured what bounds k_rollout4's MODE 3 (the mover / interact split of round 5) on BASELINE
// configs[3] (profiles/r06_interact_stream.txt): with its rare branch compiled out the step runs at the store ceiling (363 G
// env-steps/s on every layout), reading the next step's cells early and patching them in registers (one LDS round trip off the
// dependent chain, +5 VALU) made it 2.7 % SLOWER, and taking 11 VALU out of the mover (round 5) changed nothing: the step is the
// INTERACT wavefront's own instruction stream — a wavefront issues one instruction per ~5 clocks whatever the partner does
// (tools/valu_cost.hip) — plus the ~300 clocks of every rare-branch entry.  So this kernel keeps MODE 3's structure (a mover
// wavefront runs resolve_movement up to three 8-step blocks ahead and hands the interact wavefront, per step and lane, the LDS
// addresses of the two cell words the players act on) and shortens the interact wavefront's stream from ~60 to ~43 instructions:
//
//  * cell word (u32, [cell][lane]) = [K16][object][junk]: K16 = the LDS ADDRESS of the LUT row of the cell's key
//        key = 6 * type + class; type = terrain type, 7 for the pot in slot 1, 8 = "nothing" (what a player that does not
//        INTERACT acts on); class = counter: {empty, onion, tomato, dish, soup}, pot: {empty, idle 1..3, cooking, ready}
//    and a row = 5 entries (hand class) x 16 bytes.  The entry of (cell, hand) is at  word.hi16 + 16 * hand class: ONE
//    v_add_u32_sdwa (k_rollout4: four instructions — hand class, x 6, key byte, add-shift).
//  * every entry knows the class of the hand it leaves (counter classes tell what is picked up), and carries it as its
//    byte z.0 = 16 * class: the next step's address term is a byte of this step's entry, not recomputed from the hand.
//  * is_dish_pickup_useful's gate without reading the pots: one counter N = 64 x (loose dishes) - (useful pots: 1..2 idle
//    items, cooking or ready) kept by signed deltas in the entries (as the dish count was); "no loose dish and some useful
//    pot" = N < 0; the exact predicate (mdp.py:2180-2204: stale pots, live counters) is arithmetic on N in the rare branch.
//    No pot words in registers, no pot reads in the step.
//  * entry (16 bytes): .x selectors — r = v_perm(.y, pool, .x) = [new K16][new object][new hand] with
//        pool = [K16][object + add][hand] of (faced cell, hand);  .y = [new K16][delta N][dispensed object];
//        .z = [bit 7: takes a dish from the dispenser, bits 5-6: change of the full-pot count, bits 0-4: event kind][flags][add]
//             [16 * new hand class];  .w = the shaped reward float (one set of shaping rewards per batch).
//  * the pot in slot 1 has its own type, so an entry's START flag says WHICH countdown starts (no address compares).
// Semantics are k_rollout4's / the oracle's: conflict replay for player 1, stale pot states for the usefulness predicate, the
// same restart at the horizon (standard, drawn, or on a re-drawn layout).  Served batches: two players everywhere, <= 2 pots,
// <= 64 cells, one set of shaping rewards and ONE dynamics flag for the whole table (new or old), whole 256-env workgroups and
// 8-step blocks.
// ==========================================================================================
constexpr int K5_TYPES = 9, K5_POT_B = 7, K5_NOTHING = 8, K5_ROW = 80, K5_KEYS = K5_TYPES * 6;
constexpr int LUT5_BYTES = K5_KEYS * K5_ROW;  // 4 320
constexpr uint32_t Z5_PLACE = 2u << 16;
enum { F5_PLACE = 2, F5_PLATE = 4, F5_START_A = 8, F5_SERVE = 16, F5_START_B = 32, F5_CHG = 128 };
// the flags sit in byte 2 of .z: masks in place; "a dish is taken from the dispenser" is bit 31 of .z, where the sign of N meets it
constexpr uint32_t Z5_TAKE = 0x80000000u, Z5_SERVE = (uint32_t)F5_SERVE << 16, Z5_CHG = (uint32_t)F5_CHG << 16,
                   Z5_START_A = (uint32_t)F5_START_A << 16, Z5_START_B = (uint32_t)F5_START_B << 16;
constexpr uint32_t REC5_DONE = 0x80000000u, REC5_SAME = Z5_CHG;  // the mover's flag word: the horizon; both players act on one cell

template <int LUT_BASE> constexpr uint32_t k16_of(int type5, int cls) { return (uint32_t)(LUT_BASE + (type5 * 6 + cls) * K5_ROW); }

// old_dyn (mdp.py:1517-1518, 1696-1701): an INTERACT with an empty hand starts nothing; a pot that receives its third item starts
// by itself in this step's env effects — the placement's entry leaves the pot COOKING and carries the START flag
// big (grids of 65..128 cells: 16-bit cell words [key][object], key = 6 * type + class, the LUT row at 80 * key): the same entry
// with the result laid out as [0][new hand][new key][new object] and the pool as [0][hand][key][object + add]
template <int LUT_BASE>
constexpr Lut4Entry lut5_entry(int big, int old_dyn, int type5, int hc, int oc) {
    // what the new hand / object are taken from: 0 the hand, 1 the object (+ add), 4 the dispensed object, 0x0C zero
    uint32_t sel_h = 0, sel_o = 1, cobj = 0, flags = 0, add = 0, rew = RW4_NONE, hcn = (uint32_t)hc, take = 0;
    int nkey = -1, dd = 0, du = 0;  // new key (type, class) or -1 = unchanged; change of the loose-dish / useful-pot counts
    // event log (EV instances): kind = 1 + index in EVENT_TYPES (mdp.py:1027-1058) of the event this interact always logs (0:
    // none) — its USEFUL_* variant, the potting classes and useful dish pick-ups are decided at run time —, and the change of the
    // number of FULL pots (three idle items, cooking or ready: get_full_pots, mdp.py:1804-1820)
    int kind = 0, df = 0;
    const bool pot = type5 == OC_T_POT || type5 == K5_POT_B;
    if (type5 == OC_T_COUNTER) {
        if (hc == 0 && oc >= 1 && oc <= 4) {         // pick up from a counter (mdp.py:1473-1485): the class of what lies there = the hand's
            sel_h = 1; sel_o = 0x0C; nkey = 0; flags = F5_CHG; dd = oc == 3 ? -1 : 0; hcn = (uint32_t)oc;
            kind = 1 + (oc == 1 ? EV_ONION_PICKUP : oc == 2 ? EV_TOMATO_PICKUP : oc == 3 ? EV_DISH_PICKUP : EV_SOUP_PICKUP);
        } else if (hc != 0 && oc == 0) {             // drop on a counter (mdp.py:1459-1471)
            sel_h = 0x0C; sel_o = 0; nkey = hc; flags = F5_CHG; dd = hc == 3 ? 1 : 0; hcn = 0;
            kind = 1 + (hc == 1 ? EV_ONION_DROP : hc == 2 ? EV_TOMATO_DROP : hc == 3 ? EV_DISH_DROP : EV_SOUP_DROP);
        }
    } else if (type5 == OC_T_ONION_DISP) {
        if (hc == 0) { sel_h = 4; cobj = OC_O_ONION; hcn = 1; kind = 1 + EV_ONION_PICKUP; }
    } else if (type5 == OC_T_TOMATO_DISP) {
        if (hc == 0) { sel_h = 4; cobj = OC_O_TOMATO; hcn = 2; }  // (the reference logs no event here: mdp.py:1496-1498)
    } else if (type5 == OC_T_DISH_DISP) {
        if (hc == 0) { sel_h = 4; cobj = OC_O_DISH; take = Z5_TAKE; hcn = 3; kind = 1 + EV_DISH_PICKUP; }
    } else if (pot) {
        if (hc == 0 && oc >= PC_IDLE1 && oc <= PC_IDLE3 && !old_dyn) {   // begin_cooking (mdp.py:1515-1522)
            nkey = PC_COOKING; flags = F5_CHG | (type5 == K5_POT_B ? F5_START_B : F5_START_A); du = oc == PC_IDLE3 ? 1 : 0;
            df = oc == PC_IDLE3 ? 0 : 1;
        } else if (hc == 3 && oc == PC_READY) {                          // soup pickup (mdp.py:1525-1539)
            sel_h = 1; sel_o = 0x0C; nkey = PC_EMPTY; flags = F5_CHG | F5_PLATE; rew = RW4_PLATE; du = -1; hcn = 4;
            kind = 1 + EV_SOUP_PICKUP; df = -1;
        } else if ((hc == 1 || hc == 2) && oc <= PC_IDLE2) {             // add ingredient (mdp.py:1541-1568)
            sel_h = 0x0C; nkey = oc + 1; flags = F5_CHG | F5_PLACE; rew = RW4_PLACE; hcn = 0;
            kind = 1 + (hc == 1 ? EV_POTTING_ONION : EV_POTTING_TOMATO); df = oc == PC_IDLE2 ? 1 : 0;
            add = 8u + ((hc == 2 ? 1u : 0u) << oc) + (oc == 0 ? 0x80u : 0u);
            du = oc == 0 ? 1 : oc == 2 ? -1 : 0;
            if (old_dyn && oc == PC_IDLE2) {  // the third item: cooking from this step's env effects on (two idle items -> cooking: as useful as before)
                nkey = PC_COOKING; flags |= type5 == K5_POT_B ? F5_START_B : F5_START_A; du = 0;
            }
        }
    } else if (type5 == OC_T_SERVE) {
        if (hc == 4) { sel_h = 0x0C; flags = F5_SERVE; hcn = 0; kind = 1 + EV_SOUP_DELIVERY; }  // deliver (mdp.py:1570-1577)
    }
    const int dn = 64 * dd - du;
    const uint32_t z = (hcn * 16u) | (add << 8) | (flags << 16) | ((uint32_t)kind << 24) | (((uint32_t)df & 3u) << 29) | take;
    if (big) {  // pool bytes: 0 object (+ add), 1 key, 2 hand; constants (.y): byte 0 dispensed object, byte 1 new key, byte 2 delta N
        auto at = [](uint32_t sel) { return sel == 0u ? 2u : sel == 1u ? 0u : sel; };  // (4 = .y byte 0, 0x0C = zero: as they are)
        const uint32_t sel_k = nkey < 0 ? 1u : 5u;
        return Lut4Entry{at(sel_o) | (sel_k << 8) | (at(sel_h) << 16) | (0x0Cu << 24),
                         cobj | ((nkey < 0 ? 0u : (uint32_t)(type5 * 6 + nkey)) << 8) | (((uint32_t)dn & 0xFFu) << 16), z, rew};
    }
    // pool bytes: 0 hand, 1 object (+ add), 2 / 3 K16; constants (.y): byte 0 dispensed object, byte 1 delta N, bytes 2 / 3 new K16
    const uint32_t nk16 = nkey < 0 ? 0u : k16_of<LUT_BASE>(type5, nkey);
    const uint32_t sel_k = nkey < 0 ? 0x0302u : 0x0706u;
    return Lut4Entry{sel_h | (sel_o << 8) | (sel_k << 16), cobj | (((uint32_t)dn & 0xFFu) << 8) | (nk16 << 16), z, rew};
}
template <int LUT_BASE>
struct Lut5Table { Lut4Entry e[2][2][K5_KEYS][5]; };  // [big][old_dynamics][key][hand class]
template <int LUT_BASE>
constexpr Lut5Table<LUT_BASE> make_lut5() {
    Lut5Table<LUT_BASE> t{};
    for (int big = 0; big < 2; ++big)
        for (int od = 0; od < 2; ++od)
            for (int type5 = 0; type5 < K5_TYPES; ++type5)
                for (int oc = 0; oc < 6; ++oc)
                    for (int hc = 0; hc < 5; ++hc) t.e[big][od][type5 * 6 + oc][hc] = lut5_entry<LUT_BASE>(big, od, type5, hc, oc);
    return t;
}

// LDS map of k_rollout5 (one dynamic region from address 0, as k_rollout4): LUT | layout records | progress counters | ring | cells
template <bool LAY_LDS>
struct Lds5 {
    static constexpr int LUT = 0, LAY = (LUT5_BYTES + 15) & ~15, LAY_BYTES = LAY_LDS ? LDS_LAYOUT_MAX * 256 : 16;
    static constexpr int SYNC = LAY + LAY_BYTES, RING = SYNC + 64;
    // the ring, three buffers of one 8-step block each: [step][lane] pairs of LDS addresses (the cell words the two players act
    // on), then [step][lane] flag words — two arrays, so that the interact wavefront reads a record with two instructions whose
    // offsets are immediates (12-byte records: ds_read2_b32 + ds_read_b32 behind an address add)
    static constexpr int RING_XY = 8 * BLOCK * 8, RING_BUF = RING_XY + 8 * BLOCK * 4, CELLS = RING + 3 * RING_BUF;
    static_assert(LUT5_BYTES < 65536, "K16 is an LDS address in 16 bits");
};
__device__ const Lut5Table<0> g_lut5 = make_lut5<0>();

// class of what lies on a counter
__device__ __forceinline__ uint32_t counter_class5(uint32_t o) { return o == 0u ? 0u : (o & OC_O_SOUP) ? 4u : o; }
__device__ __forceinline__ uint32_t pot_type5(int k) { return k == 0 ? (uint32_t)OC_T_POT : (uint32_t)K5_POT_B; }
// (stores of a part of a cell word between whole-word reads and writes: may_alias keeps them ordered — under strict aliasing
//  the compiler moved the next step's cell reads in front of a uint16_t store)
typedef uint16_t __attribute__((may_alias)) oc_u16_alias;
// The two cell-word formats.  BIG = false: u32 [K16][object][junk], K16 = LDS address of the key's LUT row (LUT at address 0);
// BIG = true (grids of 65..128 cells): u16 [key][object]
template <bool BIG>
struct Cw5 {
    static constexpr uint32_t BYTES = BIG ? 2u : 4u, CS = (uint32_t)BLOCK * BYTES;  // CS: bytes between two cells' words in a lane's column
    static constexpr int HAND_BYTE = BIG ? 2 : 0;  // where a result word carries the hand
    static __device__ __forceinline__ uint32_t make(uint32_t type5, uint32_t cls, uint32_t obj) {
        return BIG ? (((type5 * 6u + cls) << 8) | obj) : (((type5 * 6u + cls) * K5_ROW << 16) | (obj << 8));
    }
    static __device__ __forceinline__ uint32_t rd(uint32_t a) { return BIG ? lds_rd16(a) : lds_rd32(a); }
    static __device__ __forceinline__ void wr(uint32_t a, uint32_t w) { if (BIG) *(OC_LDS oc_u16_alias*)(uintptr_t)a = (uint16_t)w; else lds_wr32(a, w); }
    static __device__ __forceinline__ uint32_t obj(uint32_t w) { return BIG ? (w & 0xFFu) : ((w >> 8) & 0xFFu); }
    static __device__ __forceinline__ uint32_t hand(uint32_t r) { return (r >> (8 * HAND_BYTE)) & 0xFFu; }
    // pot class from the word of a pot cell of type `type5` (x / 80 for x = 0, 80 .. 400)
    static __device__ __forceinline__ uint32_t pot_class(uint32_t w, uint32_t type5) {
        return BIG ? ((w >> 8) & 0xFFu) - type5 * 6u : ((((w >> 16) - type5 * 480u) * 205u) >> 14);
    }
    // the key part of a cell word alone
    static __device__ __forceinline__ void wr_key(uint32_t a, uint32_t type5, uint32_t cls) {
        if (BIG) lds_wr8(a + 1u, type5 * 6u + cls);
        else *(OC_LDS oc_u16_alias*)(uintptr_t)(a + 2u) = (uint16_t)((type5 * 6u + cls) * K5_ROW);
    }
};

// OLD: the table's layouts use old dynamics (the LUT variant above; a pot that arrives idle with three items starts in the first
//      step's env effects)
// BIG: grids of 65..128 cells (16-bit cell words, a 128-bit floor mask in the mover)
// EV: per-episode event counters (OcEventSink.d_counts / d_counts_done: env.py:382-401 game_stats) — [1 + N_EVENT_TYPES][BLOCK]
//     u32 in LDS behind the cell words for the launch (row 0 takes what is not an event), logged with ds_add
// NOOUT: the launch has no output arrays (rewards and flags both NULL: a rollout run for its final states, episode returns or event
//     counters) — the same step without the two stores (round 6: these launches used to fall to the general one-wavefront path)
template <bool LAY_LDS, bool FT8, bool OLD = false, bool BIG = false, bool EV = false, bool NOOUT = false>
__global__ __launch_bounds__(2 * BLOCK) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_rollout5(
    const OcLayout* __restrict__ g_layouts, int n_layouts, const uint16_t* layout_id, uint4* st, float4* __restrict__ rewards,
    uint8_t* __restrict__ flags, float4* __restrict__ ep_returns, int64_t n, int W, int n_obj, int horizon, uint32_t options,
    uint32_t seed_lo, uint32_t seed_hi, int64_t env_offset, int64_t t0, int n_steps, StartArgs sa, EvArgs ea) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn5[];
    using M = Lds5<LAY_LDS>;
    if ((uint32_t)(uintptr_t)(OC_LDS uint8_t*)s_dyn5 != 0u) __builtin_trap();  // folds away: the region starts at address 0
    uint4* const s_lay = reinterpret_cast<uint4*>(s_dyn5 + M::LAY);
    uint4* const s_lut = reinterpret_cast<uint4*>(s_dyn5 + M::LUT);
    constexpr int MAXP = 2;
    using CW = Cw5<BIG>;
    constexpr uint32_t CS = CW::CS;
    static_assert(M::LUT == 0, "K16 = 80 * key: the LUT starts at LDS address 0");
    const uint32_t tid = threadIdx.x & (uint32_t)(BLOCK - 1);  // lanes tid of the two halves share env e: threads 0..255 interact, the others move
    const bool mover = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
    const uint32_t blk = xcd_block();
    const int64_t e = (int64_t)blk * BLOCK + tid;  // (the host launches whole workgroups of envs only)
    Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, true, s_lay);  // contains a barrier
    {
        const uint4* src = reinterpret_cast<const uint4*>(&g_lut5) + ((BIG ? 2 : 0) + (OLD ? 1 : 0)) * (K5_KEYS * 5);
        for (int i = threadIdx.x; i < K5_KEYS * 5; i += 2 * BLOCK) {
            uint4 ent = src[i];  // one set of shaping rewards for the whole table: the entry carries the shaped reward itself
            ent.w = ent.w == RW4_PLACE ? __float_as_uint(L.rew_placement()) : ent.w == RW4_PLATE ? __float_as_uint(L.rew_soup()) : 0u;
            s_lut[i] = ent;
        }
    }
    if (threadIdx.x < 8) reinterpret_cast<uint32_t*>(s_dyn5 + M::SYNC)[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t col = (uint32_t)M::CELLS + tid * CW::BYTES;  // LDS address of this lane's column of cell words
    // two spare words per lane behind the grid: what a player that does not INTERACT acts on, then the sink of the "ready" stores of
    // pots that are not ripe (EV: and of what is not an event — the counters' row 0)
    const uint32_t noact = col + (uint32_t)n_obj * 16u * CS, dummy = noact + CS;
    LayC C = load_consts<false>(L);
    const uint32_t delta4 = make_delta4(W);
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid & ~63u));
    const uint32_t lane = tid & 63u;
    uint8_t* flg_k = flags + ((int64_t)blk * BLOCK + wave_base) * (FT8 ? 8 : 1);
    constexpr uint32_t SLOT_XY = (uint32_t)BLOCK * 8u, SLOT_F = (uint32_t)BLOCK * 4u;  // bytes of one step's records in the two arrays
    const uint32_t sync_pair = (uint32_t)M::SYNC + (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) * 8u;
    const uint32_t ring_xy = (uint32_t)M::RING + tid * 8u, ring_f = (uint32_t)(M::RING + M::RING_XY) + tid * 4u;
    auto ring_wr = [&](uint32_t buf, uint32_t k, uint32_t x, uint32_t y, uint32_t z) __attribute__((always_inline)) {
        lds_wr64(ring_xy + buf + k * SLOT_XY, x, y);
        lds_wr32(ring_f + buf + k * SLOT_F, z);
    };
    auto ring_rd = [&](uint32_t buf, uint32_t k) __attribute__((always_inline)) {
        const uint2 xy = lds_rd64(ring_xy + buf + k * SLOT_XY);
        return oc_rec3{xy.x, xy.y, lds_rd32(ring_f + buf + k * SLOT_F)};
    };
    const int n_blocks = n_steps >> 3;
    struct FloorMask { uint64_t lo, hi; };  // bit c = cell c is floor (static per layout); hi: cells 64..127 (BIG)
    auto floor_mask_of = [&](const Lay Lx) __attribute__((always_inline)) {
        FloorMask m = {0ull, 0ull};
        for (int i = 0; i < n_obj * 4; ++i) {
            const uint32_t T = Lx.u32(L_TERRAIN + 4 * i);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (((T >> (8 * b)) & 7u) == OC_T_FLOOR && (uint32_t)(4 * i + b) < Lx.u8(L_NCELLS)) {
                    if (4 * i + b < 64) m.lo |= 1ull << (4 * i + b); else m.hi |= 1ull << ((4 * i + b) & 63);
                }
        }
        return m;
    };
    auto is_floor = [](const FloorMask& m, uint32_t c) __attribute__((always_inline)) {
        uint32_t fb = (uint32_t)((BIG && c >= 64u ? m.hi : m.lo) >> (c & 63u));
        asm("" : "+v"(fb));  // (tested as a 32-bit value: k_rollout4 MODE 2)
        return (fb & 1u) != 0u;
    };

    // ---- the MOVER wavefronts: resolve_movement (mdp.py:1644-1727) for the whole launch, one 8-step block at a time, up to
    //      three blocks ahead (k_rollout4 MODE 3's mover; the flag word's shared-cell bit sits where the entries' CHG flag does)
    if (mover) {
        auto ahead = [&](uint32_t c, uint32_t d) __attribute__((always_inline)) {
            return c + (uint32_t)(int32_t)(int8_t)(uint8_t)__builtin_amdgcn_perm(0u, delta4, d);
        };
        const uint4 h = st[e];
        uint32_t P0 = h.x & 0xFFu, O0 = (h.x >> 8) & 0xFFu, P1 = h.x >> 24, O1 = h.y & 0xFFu;
        const uint32_t t_in = h.y >> 16;
        uint32_t tleft = t_in < (uint32_t)horizon ? (uint32_t)horizon - 1u - t_in : 0u;
        uint32_t over = t_in < (uint32_t)horizon ? 0u : t_in - ((uint32_t)horizon - 1u);
        FloorMask fm = floor_mask_of(L);
        uint32_t flg_off[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) flg_off[k8] = lane + (uint32_t)k8 * (uint32_t)n;
        auto produce = [&](int b, uint32_t ring) __attribute__((always_inline)) {
            if (b >= n_blocks) {  // the stub block: one "nothing" record for the last step's look-ahead, then the final pose
                ring_wr(ring, 0u, noact, noact, 0u);
                ring_wr(ring, 1u, P0 | (O0 << 8) | (P1 << 16) | (O1 << 24), tleft, over);
                return;
            }
            const Phx4 wb = philox_words(((uint64_t)t0 >> 3) + (uint64_t)b, g_lo, g_hi, seed_lo, seed_hi);
            uint32_t tile_lo = 0, tile_hi = 0;
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                uint32_t x = (k8 >> 1) == 0 ? wb.w0 : (k8 >> 1) == 1 ? wb.w1 : (k8 >> 1) == 2 ? wb.w2 : wb.w3;
                if (k8 & 1) x *= 36u;
                const uint32_t a0 = __umulhi(x, 6u), a1 = __umulhi(x * 6u, 6u);
                const uint32_t f0 = col + ahead(P0, O0) * CS, f1 = col + ahead(P1, O1) * CS;
                const uint32_t rec0 = a0 == 5u ? f0 : noact, rec1 = a1 == 5u ? f1 : noact;
                const uint32_t f_same = rec0 == rec1 ? (a0 == 5u ? REC5_SAME : 0u) : 0u;
                const uint32_t t0_ = ahead(P0, a0), t1_ = ahead(P1, a1);
                const uint32_t np0 = is_floor(fm, t0_) ? t0_ : P0, np1 = is_floor(fm, t1_) ? t1_ : P1;
                const bool collide = (np0 == np1) | ((np0 == P1) & (np1 == P0));
                const uint32_t q0 = collide ? P0 : np0, q1 = collide ? P1 : np1;
                O0 = a0 < 4u ? a0 : O0; O1 = a1 < 4u ? a1 : O1;
                P0 = q0; P1 = q1;
                uint32_t fl = 0;
                const bool done = tleft == 0u;
                tleft -= 1u;
                ring_wr(ring, (uint32_t)k8, rec0, rec1, f_same | (done ? REC5_DONE : 0u));
                if (__builtin_expect(done, 0)) {  // OvercookedEnv.step at the horizon (env.py:266-267, 321-325)
                    fl = OC_F_DONE;
                    tleft = 0u;
                    over += 1u;
                    if (options & OC_OPT_AUTO_RESET) {
                        over = 0u;
                        fl |= OC_F_RESET;
                        tleft = (uint32_t)horizon - 1u;
                        const uint32_t ep_k = sa.epoch + (uint32_t)(b * 8 + k8);
                        if (sa.enabled) {
                            if (sa.regen_count) {  // (the interact wavefront records the new id in layout_ids)
                                const uint32_t lid = draw_layout(sa, g, ep_k);
                                L = LAY_LDS ? Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u}
                                            : Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
                                fm = floor_mask_of(L);
                            }
                            const StartDraw d = draw_start(L, g, ep_k, sa.seed_lo, sa.seed_hi, sa.random_start_pos, sa.thresh);
                            P0 = d.pos0; P1 = d.pos1;
                        } else {
                            P0 = L.u8(L_START_POS); P1 = L.u8(L_START_POS + 1);
                        }
                        O0 = L.u8(L_START_OR); O1 = L.u8(L_START_OR + 1);
                    }
                }
                if (NOOUT) {
                } else if (FT8) {
                    uint32_t& half = (k8 & 4) ? tile_hi : tile_lo;
                    half = (k8 & 3) == 0 ? fl : (half | (fl << (8 * (k8 & 3))));
                } else {
                    store_flag_byte(flg_k, flg_off[k8], fl);
                }
            }
            if (FT8 && !NOOUT) {
                const uint64_t tile = ((uint64_t)tile_hi << 32) | tile_lo;
                asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(lane * 8u), "v"(tile), "s"(flg_k) : "memory");
            }
            flg_k += 8 * n;
        };
        uint32_t wbuf = 0;
        for (int j = 0; j <= n_blocks; ++j) {
            if (j >= 3)
                while (lds_poll32(sync_pair + 4u) + 2u < (uint32_t)j) __builtin_amdgcn_s_sleep(2);
            produce(j, wbuf);
            wbuf = wbuf == 2u * (uint32_t)M::RING_BUF ? 0u : wbuf + (uint32_t)M::RING_BUF;
            lds_post32(sync_pair, (uint32_t)j + 1u);
        }
        return;
    }

    // ---- the INTERACT wavefronts --------------------------------------------------------------------------------------------
    // per-lane state: hands (code in byte 0 of the last result word), the .z word of the entry that left each hand (byte 0 = 16 x
    // its class), N, the pots' countdowns (k_rollout4's: steps until ready, REM_IDLE when not cooking) and cell addresses
    uint32_t h0, h1, hz0, hz1, rem[MAXP], tk[MAXP], pa[MAXP], exotic = 0;
    int32_t N = 0;
    int32_t F = 0;  // EV: pots that are full (three idle items, cooking or ready), kept by the entries' deltas like N
    static_assert(!EV || !BIG, "the counters' row 0 is the spare cell row in front of them: 4-byte cell words");
    const uint32_t cnt0 = dummy;  // EV: this lane's column of event counters, [row][BLOCK] u32: row 0 = the spare cell word, rows 1.. behind it
    auto cnt_add = [&](uint32_t row, uint32_t v) __attribute__((always_inline)) {  // counters[row][lane] += v (row 0: nothing)
        asm volatile("ds_add_u32 %0, %1" : : "v"(cnt0 + row * (uint32_t)(BLOCK * 4)), "v"(v) : "memory");
    };
    if (EV) {
        for (int k = 0; k < N_EVENT_TYPES; ++k) lds_wr32(cnt0 + (uint32_t)(k + 1) * (BLOCK * 4u), ea.counts[e * N_EVENT_TYPES + k]);
    }
    // OLD: pots that arrive idle with three items start cooking in the coming step's env effects (mdp.py:1696-1701).  Nothing can be
    // done to such a pot in that step (every interact with a full pot is a no-op), so it is set COOKING at once, with the countdown
    // of a pot that starts in that step; only the usefulness predicate still sees it idle for one step: N gains it a step later
    int32_t N_late = 0;
    auto arrive_class = [&](uint32_t pc, uint32_t o, uint32_t& rem_k, uint32_t first_k8) __attribute__((always_inline)) {
        if (OLD && pc == PC_IDLE3) {  // ripe when rem == (index of the step in its block) + 1; a zero cook time: ready with that first step
            const uint32_t cook = cook_of(C, o);
            rem_k = max(cook, 1u) + first_k8;
            N_late += 1;
            return (uint32_t)PC_COOKING;
        }
        return pc;
    };
    auto hand_z = [](uint32_t code) __attribute__((always_inline)) { return min(code, 4u) * 16u; };
    auto pot_addrs = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k) pa[k] = (uint32_t)k < C.n_pots ? col + L.pot_cell(k) * CS : dummy;
    };
    // the cell words of an empty grid (restart) — every cell, the pots as empty pots of their slot's type
    auto write_empty_grid = [&]() __attribute__((always_inline)) {
        for (int c = 0; c < n_obj * 16; ++c) {
            const uint32_t tb = L.terrain((uint32_t)c), type = tb & 7u;
            const uint32_t type5 = (type == OC_T_POT && (tb >> 3) == 1u) ? (uint32_t)K5_POT_B : type;
            CW::wr(col + (uint32_t)c * CS, CW::make(type5, 0u, 0u));
        }
    };
    {   // load (include/oc_amd.h wire format -> key words, N, countdowns)
        const uint4 h = st[e];
        h0 = (h.x >> 16) & 0xFFu; h1 = (h.y >> 8) & 0xFFu;
        hz0 = hand_z(h0); hz1 = hand_z(h1);
        h0 <<= 8 * CW::HAND_BYTE; h1 <<= 8 * CW::HAND_BYTE;
        pot_addrs();
        int32_t dishes = 0;
        for (int p = 0; p < n_obj; ++p) {
            const uint4 v = st[(int64_t)(1 + p) * n + e];
            const uint32_t ow[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t T = L.u32(L_TERRAIN + 16 * p + 4 * q);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t o = (ow[q] >> (8 * b)) & 0xFFu, tb = (T >> (8 * b)) & 0xFFu, type = tb & 7u;
                    const uint32_t type5 = (type == OC_T_POT && (tb >> 3) == 1u) ? (uint32_t)K5_POT_B : type;
                    const uint32_t cls = type == OC_T_COUNTER ? counter_class5(o) : 0u;
                    dishes += (type == OC_T_COUNTER && o == OC_O_DISH) ? 1 : 0;
                    CW::wr(col + (uint32_t)(16 * p + 4 * q + b) * CS, CW::make(type5, cls, o));
                }
            }
        }
        CW::wr(noact, CW::make(K5_NOTHING, 0u, 0u));
        int32_t useful = 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            rem[k] = REM_IDLE; tk[k] = 0;
            if ((uint32_t)k < C.n_pots) {
                uint32_t o = CW::obj(CW::rd(pa[k]));
                const uint32_t tkb = (h.z >> (8 * k)) & 0xFFu;
                uint32_t pc = pot_class(C, o, tkb);
                if (o == OC_O_SOUP) { exotic |= 1u << k; o = 0; }  // a soup object without ingredients behaves as an empty pot
                tk[k] = tkb;
                rem[k] = pc == PC_COOKING ? cook_of(C, o) - (tkb - 1u) : REM_IDLE;
                useful += (pc != PC_EMPTY && pc != PC_IDLE3) ? 1 : 0;
                F += pc >= PC_IDLE3 ? 1 : 0;
                pc = arrive_class(pc, o, rem[k], 0u);
                CW::wr(pa[k], CW::make(pot_type5(k), pc, o));
            }
        }
        N = 64 * dishes - useful;
    }
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 epsh = {ep.z, ep.w};  // the episode's shaped returns gain the upper half of every step's reward quad (k_rollout4)
    float4* rew_k = rewards + (int64_t)blk * BLOCK;
    uint32_t rew_off[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) rew_off[k8] = (tid + (uint32_t)k8 * (uint32_t)n) * 16u;
    uint32_t step_k = 0;  // index of the step within this launch (wave-uniform): the epoch offset of a restart
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    struct Pend { uint64_t lo, hi; };  // a step's reward quad, stored by the next step of the block in the shadow of its look-ups
    auto flush = [&](const Pend& p, int k8) __attribute__((always_inline)) {
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 q = {p.lo, p.hi};
        if (NOOUT) {  // (no reward array: only the episode's shaped returns take the quad's upper half)
            asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(epsh) : "v"(p.hi));
        } else if (FT8) {  // (beside flag tiles the quads keep plain stores, beside [step][env] flag bytes they stream: k_rollout4, round 5)
            asm volatile("global_store_dwordx4 %1, %2, %4\n\tv_pk_add_f32 %0, %0, %3\n\ts_nop 0"
                         : "+v"(epsh) : "v"(rew_off[k8 & 7]), "v"(q), "v"(p.hi), "s"(rew_k) : "memory");
        } else {
            asm volatile("global_store_dwordx4 %1, %2, %4 sc1 nt\n\tv_pk_add_f32 %0, %0, %3\n\ts_nop 0"
                         : "+v"(epsh) : "v"(rew_off[k8 & 7]), "v"(q), "v"(p.hi), "s"(rew_k) : "memory");
        }
    };
    auto entry_at = [](uint32_t cw, uint32_t hz) __attribute__((always_inline)) {  // LUT address of (cell word, hand): one instruction (BIG: two)
        uint32_t a;
        if (BIG) {
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a) : "v"(cw), "v"((uint32_t)K5_ROW));
            asm("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(a) : "v"(hz));
        } else {
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0" : "=v"(a) : "v"(cw), "v"(hz));
        }
        return a;
    };
    auto interact5 = [](const uint4 ent, uint32_t h, uint32_t cw) __attribute__((always_inline)) {
        uint32_t pool = __builtin_amdgcn_perm(cw, h, BIG ? 0x0C020504u : 0x07060500u);  // [K16][object][hand] / [0][hand][key][object]
        if (BIG) asm("v_add_u32_sdwa %0, %1, %0 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(pool) : "v"(ent.z));
        else asm("v_add_u32_sdwa %0, %1, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1" : "+v"(pool) : "v"(ent.z));
        return __builtin_amdgcn_perm(ent.y, pool, ent.x);           // [new K16][new object][new hand] / [0][new hand][new key][new object]
    };
    // an entry's signed change of N (.y byte 1; BIG: byte 2)
    auto sext_b1 = [](uint32_t y) __attribute__((always_inline)) { return (int32_t)(int8_t)(uint8_t)(y >> (BIG ? 16 : 8)); };

    while (lds_poll32(sync_pair) < 1u) __builtin_amdgcn_s_sleep(1);
    uint32_t fo0, fo1, f_rec, c0, c1;
    {
        const oc_rec3 rec = ring_rd(0u, 0u);
        fo0 = rec.x; fo1 = rec.y; f_rec = rec.z;
    }
    c0 = CW::rd(fo0);
    c1 = CW::rd(fo1);
    Pend pend = {0ull, 0ull};
    // One step.  k8: its index in the block; (next_buf, next_k): the mover's record of the next step.
    auto dstep = [&](int k8, uint32_t next_buf, uint32_t next_k) __attribute__((always_inline)) {
        const uint4 e0 = lds_rd128(entry_at(c0, hz0));
        uint4 e1 = lds_rd128(entry_at(c1, hz1));
        const oc_rec3 nrec = ring_rd(next_buf, next_k);
        __builtin_amdgcn_sched_barrier(0);
        if (k8 >= 1) flush(pend, k8 - 1);  // the previous step's quad, while the look-ups are in flight
        // step_environment_effects (mdp.py:1691-1703): the countdowns; a finished one turns the pot ready — every lane stores,
        // the others into their spare word
        bool ripe[MAXP];  // (rem counts from the block's first step: the countdowns are moved once per block, not per step)
#pragma unroll
        for (int k = 0; k < MAXP; ++k) ripe[k] = rem[k] == (uint32_t)(k8 + 1);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t r0 = interact5(e0, h0, c0);
        uint32_t r1 = interact5(e1, h1, c1);
        CW::wr(fo0, r0);
        CW::wr(fo1, r1);
#pragma unroll
        for (int k = 0; k < MAXP; ++k) CW::wr_key(ripe[k] ? pa[k] : dummy, pot_type5(k), PC_READY);
        uint32_t nc0 = CW::rd(nrec.x), nc1 = CW::rd(nrec.y);  // the next step's cells: everything this step writes has been issued
        const int32_t N_mid = N + sext_b1(e0.y);
        int32_t N_new = N_mid + sext_b1(e1.y);
        int32_t late_now = 0;  // OLD: pots that arrived idle and full count as useful from the step after their first one on
        if (OLD) { late_now = N_late; N_late = 0; }
        // ---- ONE branch for everything rare: cooking starts, deliveries, dish pick-ups that may be useful (N < 0 before or
        //      after player 0's interact: no loose dish, some useful pot), a shared cell player 0 has changed, the horizon
        const uint32_t gate = ((uint32_t)(N | N_mid) & Z5_TAKE) | Z5_SERVE | Z5_START_A | Z5_START_B | (EV ? Z5_PLACE : 0u);
        uint32_t c1_seen = c1;  // EV: the cell word player 1's interact saw (player 0's result when it redoes it)
        const uint32_t rare_bits = ((e0.z | e1.z) & gate) | ((e0.z | REC5_DONE) & f_rec);
        uint64_t q_lo, q_hi;  // the reward quad as two register pairs: zeros, and the two entries' shaped floats
        {
            const uint64_t a64 = ((uint64_t)e0.w << 32) | e0.z, b64 = ((uint64_t)e1.w << 32) | e1.z;
            asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(q_hi) : "v"(a64), "v"(b64));
            asm("v_mov_b64 %0, 0" : "=v"(q_lo));
        }
        uint32_t nh0 = r0, nh1 = r1, nz0 = e0.z, nz1 = e1.z;
        int32_t F_reset = -1;  // EV: the full pots of a new episode's start state
        if (__builtin_expect(rare_bits != 0u, 0)) {
            bool grid_changed = false;
            const uint32_t hb0 = CW::hand(h0), hb1 = CW::hand(h1), hn0 = CW::hand(r0);
            if (e0.z & f_rec & Z5_CHG) {  // player 1 acts on the cell player 0 has just changed: redo its interact on what is there now (Q2 / Q3)
                e1 = lds_rd128(entry_at(r0, hz1));
                r1 = interact5(e1, h1, r0);
                c1_seen = r0;
                nh1 = r1; nz1 = e1.z;
                CW::wr(fo1, r1);
                N_new = N_mid + sext_b1(e1.y);
                grid_changed = true;
            }
            const uint32_t fz = e0.z | e1.z;
            // begin_cooking (mdp.py:1515-1522) = load the countdown: tick 0 now, cooked once by this step's env effects (Q4)
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                const uint32_t sk = k == 0 ? Z5_START_A : Z5_START_B;
                if (fz & sk) {
                    const uint32_t soup = CW::obj((e0.z & sk) ? r0 : r1);
                    const uint32_t cook = cook_of(C, soup);
                    rem[k] = cook + (uint32_t)k8;  // (= cook - 1 steps after this one, counted from the block's first step)
                    exotic &= ~(1u << k);
                    if (cook <= 1u) {  // ready with this step's env effects (cook == 0: at once, never ticks)
                        CW::wr_key(pa[k], pot_type5(k), PC_READY);
                        grid_changed = true;
                    }
                }
            }
            float4 rw = make_float4(0.f, 0.f, __uint_as_float(e0.w), __uint_as_float(e1.w));
            if (fz & (Z5_SERVE | Z5_TAKE)) {
                // is_dish_pickup_useful (mdp.py:2180-2204): pot_states of before the interacts (the useful pots in N), live hands
                // and counters; N = 64 x dishes - useful pots with at most 2 pots
                const int32_t dishes_b = (N + 63) >> 6, useful = 64 * dishes_b - N;
                const bool du0 = (((hb1 == OC_O_DISH) ? 1 : 0) < useful) & (dishes_b == 0);
                const bool du1 = (((hn0 == OC_O_DISH) ? 1 : 0) < useful) & (N_mid <= 0);
                float4 r;
                r.z = (((e0.z & Z5_TAKE) != 0u) & du0) ? C.rew_dish : 0.f;
                r.w = (((e1.z & Z5_TAKE) != 0u) & du1) ? C.rew_dish : 0.f;
                r.x = r.y = 0.f;
                if (fz & Z5_SERVE) {  // deliver_soup (mdp.py:1631-1642)
                    r.x = (e0.z & Z5_SERVE) ? L.value(recipe_idx(hb0) & 15u) : 0.f;
                    r.y = (e1.z & Z5_SERVE) ? L.value(recipe_idx(hb1) & 15u) : 0.f;
                }
                ep.x += r.x; ep.y += r.y;
                rw.x = r.x; rw.y = r.y; rw.z += r.z; rw.w += r.w;
                if (EV) {  // useful_dish_pickup (only a dish from the dispenser can be one: a dish on a counter is a loose dish)
                    cnt_add((((e0.z & Z5_TAKE) != 0u) & du0) ? 1u + EV_USEFUL_DISH_PICKUP : 0u, 1u);
                    cnt_add((((e1.z & Z5_TAKE) != 0u) & du1) ? 1u + EV_USEFUL_DISH_PICKUP : 0u, 1u << 16);
                }
            }
            if (EV && (fz & Z5_PLACE)) {  // log_object_potting (mdp.py:2251-2308): the classes of (old soup, ingredient), as interact_events
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const uint32_t ez = pl ? e1.z : e0.z, o = CW::obj(pl ? c1_seen : c0), hb = pl ? hb1 : hb0;
                    const uint32_t n = (o >> 3) & 3u, nt = __popc(o & 7u), no = n - nt, pi = no + 3u * nt;
                    const uint32_t pw = pi < 4u ? C.pclass[0] : C.pclass[1], tom = hb == OC_O_TOMATO ? 1u : 0u;
                    const uint32_t nib = (ez & Z5_PLACE) ? (pw >> (8u * (pi & 3u) + 4u * tom)) & 0xFu : 0u;
                    const uint32_t v = pl ? 1u << 16 : 1u;
                    cnt_add((nib & 1u) ? 1u + EV_OPTIMAL_ONION_POTTING + tom : 0u, v);
                    cnt_add((nib & 2u) ? 1u + EV_VIABLE_ONION_POTTING + tom : 0u, v);
                    cnt_add((nib & 4u) ? 1u + EV_CATASTROPHIC_ONION_POTTING + tom : 0u, v);
                    cnt_add((nib & 8u) ? 1u + EV_USELESS_ONION_POTTING + tom : 0u, v);
                }
            }
            if (f_rec & REC5_DONE) {  // OvercookedEnv.step at the horizon (env.py:266-267, 321-325); the mover stores the flag byte
                if (options & OC_OPT_AUTO_RESET) {
                    StartDraw d;
                    d.held0 = d.held1 = d.ticks0 = d.ticks1 = d.pots0 = d.pots1 = 0u;
                    if (sa.enabled) {  // the batch's start_state_fn, drawn from (seed, global env, epoch of this step)
                        if (sa.regen_count) {  // ... on a layout drawn for the new episode (regen_mdp, env.py:288-302)
                            const uint32_t lid = draw_layout(sa, g, sa.epoch + step_k);
                            sa.layout_ids[e] = (uint16_t)lid;
                            L = LAY_LDS ? Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u}
                                        : Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
                            C = load_consts<false>(L);
                            pot_addrs();
                        }
                        d = draw_start(L, g, sa.epoch + step_k, sa.seed_lo, sa.seed_hi, sa.random_start_pos, sa.thresh);
                    }
                    write_empty_grid();
                    int32_t useful = 0;
                    exotic = 0;
                    F_reset = 0;
#pragma unroll
                    for (int k = 0; k < MAXP; ++k) {
                        rem[k] = REM_IDLE; tk[k] = 0;
                        if ((uint32_t)k < C.n_pots) {
                            const uint32_t o = d.pot_obj((uint32_t)k), tkb = d.tick((uint32_t)k);
                            uint32_t pc = pot_class(C, o, tkb);
                            F_reset += pc >= PC_IDLE3 ? 1 : 0;
                            tk[k] = tkb;
                            rem[k] = pc == PC_COOKING ? cook_of(C, o) - (tkb - 1u) + (uint32_t)(k8 + 1) : REM_IDLE;
                            useful += (pc != PC_EMPTY && pc != PC_IDLE3) ? 1 : 0;
                            pc = arrive_class(pc, o, rem[k], (uint32_t)(k8 + 1));
                            CW::wr(pa[k], CW::make(pot_type5(k), pc, o));
                        }
                    }
                    N_new = -useful;
                    late_now = 0;  // (pots drawn idle and full for the new episode: N_late, from the next step on)
                    nh0 = d.held0 << (8 * CW::HAND_BYTE); nh1 = d.held1 << (8 * CW::HAND_BYTE);
                    nz0 = hand_z(d.held0); nz1 = hand_z(d.held1);
                    ep = zero4;  // the episode ends with this step: its returns restart from zero
                    epsh.x = -rw.z; epsh.y = -rw.w;
                    grid_changed = true;
                }
            }
            if (grid_changed) {  // read the next step's cells again, behind everything this step wrote
                nc0 = CW::rd(nrec.x);
                nc1 = CW::rd(nrec.y);
            }
            q_lo = ((uint64_t)__float_as_uint(rw.y) << 32) | __float_as_uint(rw.x);
            q_hi = ((uint64_t)__float_as_uint(rw.w) << 32) | __float_as_uint(rw.z);
            // (wait for the reads in here: left pending, the join behind the branch would wait for them at its first LDS use)
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        if (EV) {
            // event_infos of the step (mdp.py:2121-2308) into the episode's counters: the event each entry always logs, then its
            // USEFUL_* variant (is_ingredient_pickup_useful / is_ingredient_drop_useful / is_dish_drop_useful, mdp.py:2206-2249:
            // the full pots of BEFORE the interacts, the other player's hand as it is when the player acts — player 0's new
            // hand for player 1).  Pick-ups are useful unless every pot is full and the other player holds no dish, drops of
            // ingredients exactly then: one of the two masks is on.
            constexpr uint32_t PICKM = (1u << (1 + EV_TOMATO_PICKUP)) | (1u << (1 + EV_ONION_PICKUP));
            constexpr uint32_t DROPM = (1u << (1 + EV_TOMATO_DROP)) | (1u << (1 + EV_ONION_DROP)), DDM = 1u << (1 + EV_DISH_DROP);
            const uint32_t k0 = (e0.z >> 24) & 31u, k1 = (e1.z >> 24) & 31u;
            cnt_add(k0, 1u);
            cnt_add(k1, 1u << 16);
            const bool all_full = F == (int32_t)C.n_pots, none_full = F == 0;
            auto useful_row = [&](uint32_t kind, uint32_t other_z) __attribute__((always_inline)) {
                const uint32_t oc16 = other_z & 0xFFu;  // 16 x the class of the other player's hand: 48 a dish, 16 an onion
                const uint32_t m = ((all_full & (oc16 != 48u)) ? DROPM : PICKM) | ((none_full & (oc16 != 16u)) ? DDM : 0u);
                return ((m >> kind) & 1u) ? kind + 1u : 0u;
            };
            cnt_add(useful_row(k0, hz1), 1u);
            cnt_add(useful_row(k1, e0.z), 1u << 16);
            F += ((int32_t)(e0.z << 1) >> 30) + ((int32_t)(e1.z << 1) >> 30);
            if (F_reset >= 0) F = F_reset;
            if (f_rec & REC5_DONE) {  // the episode ends with this step: publish its counts, start the next one from zero
                for (int k = 0; k < N_EVENT_TYPES; ++k) {
                    const uint32_t a = cnt0 + (uint32_t)(k + 1) * (BLOCK * 4u);
                    if (ea.counts_done) ea.counts_done[e * N_EVENT_TYPES + k] = lds_rd32(a);
                    if ((options & OC_OPT_AUTO_RESET) || ea.clear_on_done) lds_wr32(a, 0u);
                }
            }
        }
        if (k8 < 7) { pend.lo = q_lo; pend.hi = q_hi; }
        else { const Pend now = {q_lo, q_hi}; flush(now, 7); }
        h0 = nh0; h1 = nh1; hz0 = nz0; hz1 = nz1; N = N_new - late_now;
        fo0 = nrec.x; fo1 = nrec.y; f_rec = nrec.z; c0 = nc0; c1 = nc1;
        step_k += 1u;
    };
    uint32_t rbuf = 0;  // (wave-uniform) offset of the ring buffer that holds the block being run
    for (int b = 0; b < n_blocks; ++b) {
        // Block b's own records are there (checked by block b - 1's step 7); step 7 looks ahead to the first record of block
        // b + 1: the mover must have finished b + 2 blocks by then (the stub behind the launch counts as one).  The count is read
        // before step 6 and looked at after it, so that the read's latency is not the loop's.
        const uint32_t cur = rbuf;
        rbuf = rbuf == 2u * (uint32_t)M::RING_BUF ? 0u : rbuf + (uint32_t)M::RING_BUF;
#pragma unroll
        for (int k8 = 0; k8 < 6; ++k8) dstep(k8, cur, (uint32_t)(k8 + 1));
        const uint32_t produced = *(const volatile OC_LDS uint32_t*)(uintptr_t)sync_pair;
        dstep(6, cur, 7u);
        if (__builtin_amdgcn_readfirstlane((int)produced) < b + 2)
            while (lds_poll32(sync_pair) < (uint32_t)b + 2u) __builtin_amdgcn_s_sleep(1);
        dstep(7, rbuf, 0u);
#pragma unroll
        for (int k = 0; k < MAXP; ++k) rem[k] -= 8u;
        rew_k += 8 * n;
        lds_post32(sync_pair + 4u, (uint32_t)b + 1u);
    }
    // ---- store (key words -> wire format); the pose and the episode clock come from the mover's stub record
    const oc_rec3 fin = ring_rd(rbuf, 1u);
    {
        const uint32_t t = min((uint32_t)horizon - 1u - fin.y + fin.z, 0xFFFFu);  // the wire format's u16: saturates
        uint4 h;
        h.x = (fin.x & 0xFFFFu) | (CW::hand(h0) << 16) | ((fin.x >> 16) << 24);
        h.y = (fin.x >> 24) | (CW::hand(h1) << 8) | (t << 16);
        h.z = 0; h.w = 0;
        uint32_t fix_cell[MAXP], fix_obj[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            fix_cell[k] = 0xFFFFFFFFu; fix_obj[k] = 0;
            if ((uint32_t)k < C.n_pots) {
                const uint32_t cw = CW::rd(pa[k]), o = CW::obj(cw), pc = CW::pot_class(cw, pot_type5(k));
                uint32_t tkb = tk[k];
                const bool live = rem_live(rem[k]);
                if (live) {
                    const uint32_t cook = cook_of(C, o);
                    tkb = (pc == PC_COOKING ? cook - rem[k] : cook) + 1u;
                }
                if (pc < PC_COOKING) tkb = 0;  // empty or idle
                h.z |= tkb << (8 * k);
                if (o == 0u && ((exotic >> k) & 1u) && !live) { fix_cell[k] = L.pot_cell(k); fix_obj[k] = OC_O_SOUP; }
            }
        }
        st[e] = h;
        for (int p = 0; p < n_obj; ++p) {
            uint32_t ow[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ow[q] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t c = (uint32_t)(16 * p + 4 * q + b);
                    uint32_t o = CW::obj(CW::rd(col + c * CS));
#pragma unroll
                    for (int k = 0; k < MAXP; ++k) o = c == fix_cell[k] ? fix_obj[k] : o;
                    ow[q] |= o << (8 * b);
                }
            }
            st[(int64_t)(1 + p) * n + e] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
    if (EV)
        for (int k = 0; k < N_EVENT_TYPES; ++k) ea.counts[e * N_EVENT_TYPES + k] = lds_rd32(cnt0 + (uint32_t)(k + 1) * (BLOCK * 4u));
    ep.z = epsh.x; ep.w = epsh.y;
    if (ep_returns) ep_returns[e] = ep;
}
