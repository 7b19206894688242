// rollout_encode.hpp — K transitions WITH the lossless observation of every step: k_rollout_encode
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, after step_table.hpp and encode.hpp.
#pragma once

// ------------------------------------------------------------------------------------------
// k_rollout_encode: BASELINE configs[2] as SURVEY.md 8d-3 states it — the rollout of configs[1] (in-kernel Philox
// actions, auto-reset, rewards / flags every step) plus lossless_state_encoding (mdp.py:2385-2561) of every env after
// every step — for a batch with ONE layout, u8 or f32 observations, random policy or caller-supplied actions.
//
// One launch of a one-step kernel is a ~6 us latency chain and the observation kernel has ~6 us of fill time of its
// own; stepping inside the persistent observation kernel once per 19-env group was measured slower.
// Here the roles are the other way round: the env stays with its lane for all K steps exactly as in the fused rollout kernels
// (registers + [cell][lane] words in LDS), and after each step every WAVEFRONT encodes its own 64 envs by itself:
//   * the wire-format header of each env (players, timestep, pot ticks) goes to a 16-byte LDS slot, the pots' soup
//     codes back into their cell words — then any lane can read any env of its wavefront;
//   * per sub-group of G envs: copy the static-layer template into the wavefront's private LDS image (16-byte moves),
//     scatter the dynamic values (players and objects in separate loops, branch-free layer writes: a wavefront executes
//     the union of its lanes' paths), stream the image to its place in obs[step] as contiguous 16-byte stores.
// No workgroup barrier between the phases: LDS operations of one wavefront execute in order, so they need only a
// compiler-level fence.  The transition costs ~1 300 clk per step; copy + scatter of a 9x5 wavefront ~22 us without the
// output stores, just under the ~28 us HBM needs for the 153 MB of a step — the loop runs at the encoder's write rate.
// NW = 8 (small grids, where four wavefronts cannot keep up with HBM) adds four helper wavefronts that own no env and
// encode every other sub-group, between two workgroup barriers per step.
// What lies on the grid is scattered from a COMPACT LIST (round 4): right after its transition every lane walks its own
// env's object bytes once (lane = env: all 64 lanes busy) and leaves (cell | object << 8) entries in LDS, at most
// RE_LIST_CAP per env, plus the sub-group's largest count; the sub-group's scatter then runs lane = (env, k-th object) over
// next_pow2(largest count) slots per env instead of lane = (env, object dword) over every dword of the grid, most of which
// are empty (9x5: 3 x 16 slots per four envs, a divergent loop over the bytes of each).  A sub-group with an env above the
// cap takes the dword loop.  Cell -> item offsets come from a 64-entry table instead of a divide by W.
// ------------------------------------------------------------------------------------------
constexpr int RE_LIST_CAP = 14;  // (what fits next to four 12-env images of a 9x5 grid in 160 KB)
// dynamic LDS behind the cell words, the template and the headers: the lists, their counts (include in the carve of oc_rollout_encode)
constexpr size_t RE_LIST_BYTES = (size_t)RE_LIST_CAP * BLOCK * sizeof(uint16_t) + BLOCK;
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MAXP, int FAST, typename T, int NW>
__global__ __launch_bounds__(NW * 64) void k_rollout_encode(const OcLayout* __restrict__ g_layouts, uint4* st,
                                                          const uint8_t* __restrict__ actions,
                                                          float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                          float4* __restrict__ ep_returns, uint8_t* __restrict__ obs_bytes,
                                                          int64_t obs_step_stride, int64_t n, int W, int H, int n_obj,
                                                          int horizon, uint32_t options, uint32_t seed_lo,
                                                          uint32_t seed_hi, int64_t env_offset, int64_t t0, int n_steps,
                                                          int unit, int group_envs, StartArgs sa) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_cells3[];  // [n_obj * 16][BLOCK], then template / headers / images
    __shared__ uint4 s_lay[16];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    __shared__ uint8_t s_move[FAST == 3 ? 64 * 8 : 8];
    __shared__ uint64_t s_urgent[4];  // (NW = 8) per owner wavefront: which of its envs are in their last 40 steps
    __shared__ uint16_t s_ioff[64];   // cell -> offset of its item in a view, in values: 26 * (x * H + y)
    __shared__ uint32_t s_gmax[4][64];  // per owner wavefront and sub-group: the largest object count of its envs
    static_assert(NW == 4 || NW == 8, "four owner wavefronts, optionally four helpers");
    const int cells_n = W * H;
    const int items_per_env = 2 * cells_n;
    const size_t env_bytes = (size_t)items_per_env * OC_NUM_LAYERS * sizeof(T);
    const int unit_chunks = (int)(env_bytes * unit / 16);       // the template: `unit` envs, a multiple of 16 bytes
    const int img_chunks = unit_chunks * (group_envs / unit);   // one wavefront's image: group_envs envs
    uint4* s_tmpl = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(s_cells3) + (size_t)n_obj * 16 * BLOCK * sizeof(uint16_t));
    uint4* s_hdr = s_tmpl + unit_chunks;                        // [BLOCK] wire-format plane 0 of each env
    uint16_t* s_list = reinterpret_cast<uint16_t*>(s_hdr + BLOCK);  // [RE_LIST_CAP][BLOCK] cell | object << 8, in cell order
    uint8_t* s_cnt = reinterpret_cast<uint8_t*>(s_list + RE_LIST_CAP * BLOCK);  // [BLOCK] objects on the env's grid (255 = more)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* img = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(s_hdr + BLOCK) + RE_LIST_BYTES) + (size_t)wave * img_chunks;
    T* imgT = reinterpret_cast<T*>(img);
    T* tmpl = reinterpret_cast<T*>(s_tmpl);
    const uint32_t inv_w = 65536u / (uint32_t)W + 1u;

    // NW = 8: lanes 256..511 are four helper wavefronts.  They own no env; wavefront 4 + w encodes every other sub-group
    // of owner wavefront w's envs (between two workgroup barriers per step), halving what a wavefront does per step.
    const bool owner = NW == 4 || threadIdx.x < BLOCK;
    const int ow = wave & 3, part = wave >> 2;
    const uint32_t blk = xcd_block();  // (common.hpp: each XCD owns a contiguous eighth of the envs — and of every step's observations)
    const int64_t e = (int64_t)blk * BLOCK + (threadIdx.x & (BLOCK - 1));
    const bool active = owner && e < n;
    // caller actions: the first step's are requested before the tables are staged, step k + 1's while step k is encoded
    uint32_t a01_next = (actions && active && n_steps > 0) ? reinterpret_cast<const uint16_t*>(actions)[e] : 0u;
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    for (int i = threadIdx.x; i < unit_chunks; i += BLOCK) s_tmpl[i] = make_uint4(0, 0, 0, 0);
    const Lay L = stage_layouts<true>(g_layouts, 1, nullptr, e, active, s_lay);  // contains the barrier
    if (threadIdx.x < 64) {
        const uint32_t c = threadIdx.x, y = (c * inv_w) >> 16, x = c - y * (uint32_t)W;
        s_ioff[c] = (uint16_t)((x * (uint32_t)H + y) * OC_NUM_LAYERS);  // (cells beyond the grid are never looked up)
    }
    if (FAST == 3) {  // MOVE[cell * 8 + action] for the batch's single layout (at most 64 cells)
        const int nc = (int)L.u8(L_NCELLS);
        for (int i = threadIdx.x; i < nc * 8; i += BLOCK) {
            const int c = i >> 3, a = i & 7;
            int t = c;
            if (a < 4) {
                const int t2 = c + (a == 0 ? -W : a == 1 ? W : a == 2 ? 1 : -1);
                if (t2 >= 0 && t2 < nc && (L.terrain((uint32_t)t2) & 7u) == OC_T_FLOOR) t = t2;
            }
            s_move[i] = (uint8_t)t;
        }
    }
    // static terrain layers (mdp.py:2449-2465) of `unit` envs, both views
    for (int q = threadIdx.x; q < unit * cells_n; q += BLOCK) {
        const int u = q / cells_n;
        const uint32_t c = (uint32_t)(q - u * cells_n);
        const uint32_t type = L.terrain(c) & 7u;
        if (type != OC_T_FLOOR) {
            const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W, i = x * (uint32_t)H + y;
            const uint32_t layer = (0x0F0E0A0D0C0B00ull >> (8u * type)) & 0xFFu;  // P->10 X->11 O->12 T->13 D->14 S->15
            T* base = tmpl + (size_t)u * items_per_env * OC_NUM_LAYERS;
            base[((size_t)i) * OC_NUM_LAYERS + layer] = (T)1;
            base[((size_t)cells_n + i) * OC_NUM_LAYERS + layer] = (T)1;
        }
    }
    __syncthreads();  // NW = 4: the last workgroup barrier, from here on every wavefront runs by itself
    const int64_t wave_e0 = (int64_t)blk * BLOCK + (int64_t)ow * 64;
    const int n_wave = (int)max((int64_t)0, min((int64_t)64, n - wave_e0));  // envs of the owner wavefront
    if (NW == 4 && n_wave == 0) return;

    uint16_t* cells = s_cells3 + (threadIdx.x & (BLOCK - 1));
    const LayC C = load_consts<true>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    const uint32_t delta4 = make_delta4(W);
    Env3<MAXP> s;
    float4 ep = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        load_env3<MAXP>(C, L, st, n, e, n_obj, s, cells);
        if (ep_returns) ep = ep_returns[e];
    }
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    uint32_t rnd[4] = {0, 0, 0, 0};
    const int obj_dwords = n_obj * 4;
    const uint16_t* wcells = s_cells3 + ow * 64;  // cell c of the owner wavefront's env l: wcells[c * BLOCK + l]
    const uint4* whdr = s_hdr + ow * 64;

    const int my_group = lane / group_envs;  // the sub-group this lane's env is encoded with
    for (int k = 0; k < n_steps; ++k) {
        // ---- the transition (get_state_transition + OvercookedEnv.step bookkeeping), as k_step3 does it
        bool urgent = false;
        if (active) {
            uint32_t a0, a1;
            if (actions) {
                const uint32_t a01 = a01_next;
                if (k + 1 < n_steps) a01_next = reinterpret_cast<const uint16_t*>(actions)[(int64_t)(k + 1) * n + e];
                a0 = a01 & 0xFFu; a1 = a01 >> 8;
            } else {
                const uint64_t t = (uint64_t)(t0 + k);
                const uint32_t s8 = (uint32_t)t & 7u;
                if (k == 0 || s8 == 0u) {
                    const uint64_t blk = t >> 3;
                    philox4x32_10((uint32_t)blk, g_lo, g_hi, (uint32_t)(blk >> 32), seed_lo, seed_hi, rnd);
                }
                draw_actions(rnd, s8, a0, a1);
            }
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t fl;
            if (__builtin_expect(a0 > 5u || a1 > 5u, 0)) {
                fl = OC_F_BAD_ACTION;  // mdp.py:1394-1398 raises: the env stays untouched
            } else {
                env_step3<MAXP, FAST>(C, L, lut, cells, s, delta4, a0, a1, r, 0ull, s_move);
                fl = finish_step3<MAXP>(C, L, n_obj, cells, s, horizon, options, r, ep, sa, g, sa.epoch + (uint32_t)k);
            }
            if (rewards) rewards[(int64_t)k * n + e] = r;
            if (flags) flags[(int64_t)k * n + e] = (uint8_t)fl;
            // what the observation needs from this lane's registers: the wire header, the pots' soup codes in the grid
            uint4 h;
            h.x = s.pos0 | (s.or0 << 8) | (s.held0 << 16) | (s.pos1 << 24);
            h.y = s.or1 | (s.held1 << 8) | (min(s.t, 0xFFFFu) << 16);
            h.z = 0; h.w = 0;
#pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                if ((uint32_t)p < C.n_pots) {
                    wr_obj3(cells, L.pot_cell(p), s.ps[p]);
                    h.z |= s.tk[p] << (8 * (p & 3));
                }
            }
            s_hdr[threadIdx.x] = h;
            urgent = (horizon - (int)s.t) < 40;
        }
        // the compact list of what lies on this lane's grid, and the largest count of each sub-group
        if (owner) {
            s_gmax[ow][lane] = 0u;
            wave_fence();
            if (active) {
                uint32_t cnt = 0;
                for (int j = 0; j < obj_dwords; ++j) {
                    const uint32_t c0 = cells[(4 * j + 0) * BLOCK], c1 = cells[(4 * j + 1) * BLOCK];
                    const uint32_t c2 = cells[(4 * j + 2) * BLOCK], c3 = cells[(4 * j + 3) * BLOCK];
                    uint32_t w = (c0 & 0xFFu) | ((c1 & 0xFFu) << 8) | ((c2 & 0xFFu) << 16) | (c3 << 24);  // the four object bytes
                    while (w != 0u) {
                        const uint32_t b4 = (uint32_t)(__ffs((int)w) - 1) >> 3;
                        const uint32_t o = (w >> (8u * b4)) & 0xFFu;
                        w &= ~(0xFFu << (8u * b4));
                        if (cnt < (uint32_t)RE_LIST_CAP) s_list[cnt * BLOCK + threadIdx.x] = (uint16_t)((4u * (uint32_t)j + b4) | (o << 8));
                        ++cnt;
                    }
                }
                s_cnt[threadIdx.x] = (uint8_t)min(cnt, 255u);
                atomicMax(&s_gmax[ow][my_group], cnt);
            }
        }
        uint64_t urgent_mask = __ballot(urgent);
        if (NW == 8) {
            if (owner && lane == 0) s_urgent[ow] = urgent_mask;
            __syncthreads();  // the step's cell words and headers are in LDS for the helpers
            urgent_mask = s_urgent[ow];
        } else {
            wave_fence();
        }

        // ---- lossless_state_encoding of this wavefront's envs, G at a time through its private LDS image
        uint8_t* obs_k = obs_bytes + (int64_t)k * obs_step_stride;
        for (int l0 = part * group_envs; l0 < n_wave; l0 += (NW / 4) * group_envs) {
            const int ne = min(group_envs, n_wave - l0);
            const int n_units = (ne + unit - 1) / unit;
            for (int i = lane; i < unit_chunks; i += 64) {
                const uint4 v = s_tmpl[i];
                for (int u = 0; u < n_units; ++u) img[(size_t)u * unit_chunks + i] = v;
            }
            wave_fence();
            // dynamic values.  A wavefront executes the union of its lanes' paths, so the two kinds of task run in
            // separate loops and the layer writes of an object are branch-free (enc_object_writes).
            // players (mdp.py:2468-2479, ordering 2423-2434) and what they hold: lane = (env, player)
            for (int t = lane; t < 2 * ne; t += 64) {
                const int le = t >> 1, pl = t & 1;
                const uint4 hw = whdr[l0 + le];
                const uint32_t pos = pl == 0 ? (hw.x & 0xFFu) : (hw.x >> 24);
                const uint32_t ori = pl == 0 ? ((hw.x >> 8) & 0xFFu) : (hw.y & 0xFFu);
                const uint32_t held = pl == 0 ? ((hw.x >> 16) & 0xFFu) : ((hw.y >> 8) & 0xFFu);
                if (pos != 0xFFu) {
                    const uint32_t io = s_ioff[pos & 63u];
                    T* own = imgT + ((uint32_t)le * (uint32_t)items_per_env + (uint32_t)pl * (uint32_t)cells_n) * OC_NUM_LAYERS + io;        // view pl
                    T* other = imgT + ((uint32_t)le * (uint32_t)items_per_env + (uint32_t)(1 - pl) * (uint32_t)cells_n) * OC_NUM_LAYERS + io;  // the other view
                    own[0] = (T)1; own[2 + ori] = (T)1;
                    other[1] = (T)1; other[6 + ori] = (T)1;
                    if (held) { enc_object_writes<T>(own, held, false, 0u, 0u); enc_object_writes<T>(other, held, false, 0u, 0u); }
                }
            }
            // objects on the grid (mdp.py:2482-2534) from the compact lists: lane = (env, k-th object of its grid)
            const uint32_t kmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_gmax[ow][l0 / group_envs]);
            if (kmax <= (uint32_t)RE_LIST_CAP) {
                if (kmax != 0u) {
                    const int sh = kmax <= 1u ? 0 : 32 - __builtin_clz(kmax - 1u);  // slots per env: the next power of two
                    const int tasks = ne << sh;
                    for (int t = lane; t < tasks; t += 64) {
                        const int le = t >> sh, l = l0 + le;
                        const uint32_t k = (uint32_t)t & ((1u << sh) - 1u);
                        if (k < (uint32_t)s_cnt[ow * 64 + l]) {
                            const uint32_t ent = s_list[k * BLOCK + ow * 64 + l];
                            const uint32_t c = ent & 0xFFu, o = ent >> 8;
                            const uint32_t tkw = whdr[l].z;
                            const uint32_t tc = L.terrain(c);
                            const bool in_pot = (tc & 7u) == OC_T_POT;
                            const uint32_t tk = (tkw >> (8u * ((tc >> 3) & 3u))) & 0xFFu;
                            const uint32_t ct = L.cook_time(recipe_idx(o) & 15u);
                            T* item = imgT + (uint32_t)le * (uint32_t)items_per_env * OC_NUM_LAYERS + s_ioff[c & 63u];
                            enc_object_writes<T>(item, o, in_pot, tk, ct);
                            enc_object_writes<T>(item + (uint32_t)cells_n * OC_NUM_LAYERS, o, in_pot, tk, ct);
                        }
                    }
                }
            } else {  // an env of the sub-group holds more objects than a list takes: lane = (env, object dword), 8 or 16 slots per env
                const int sh = obj_dwords <= 8 ? 3 : 4;
                for (int le = lane >> sh; le < ne; le += (64 >> sh)) {
                    const int j = lane & ((1 << sh) - 1);
                    if (j < obj_dwords) {
                        const int l = l0 + le;
                        const uint32_t c0 = wcells[(4 * j + 0) * BLOCK + l], c1 = wcells[(4 * j + 1) * BLOCK + l];
                        const uint32_t c2 = wcells[(4 * j + 2) * BLOCK + l], c3 = wcells[(4 * j + 3) * BLOCK + l];
                        uint32_t w = (c0 & 0xFFu) | ((c1 & 0xFFu) << 8) | ((c2 & 0xFFu) << 16) | (c3 << 24);  // the four object bytes
                        if (w != 0u) {
                            const uint32_t tkw = whdr[l].z;
                            T* env_img = imgT + (size_t)le * items_per_env * OC_NUM_LAYERS;
                            while (w != 0u) {
                                const uint32_t b4 = (uint32_t)(__ffs((int)w) - 1) >> 3;  // lowest non-empty cell of the dword
                                const uint32_t o = (w >> (8u * b4)) & 0xFFu;
                                w &= ~(0xFFu << (8u * b4));
                                const uint32_t c = 4u * (uint32_t)j + b4;
                                const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W, i = x * (uint32_t)H + y;
                                const uint32_t tc = L.terrain(c);
                                const bool in_pot = (tc & 7u) == OC_T_POT;
                                const uint32_t tk = (tkw >> (8u * ((tc >> 3) & 3u))) & 0xFFu;
                                const uint32_t ct = L.cook_time(recipe_idx(o) & 15u);
                                enc_object_writes<T>(env_img + (size_t)i * OC_NUM_LAYERS, o, in_pot, tk, ct);
                                enc_object_writes<T>(env_img + ((size_t)cells_n + i) * OC_NUM_LAYERS, o, in_pot, tk, ct);
                            }
                        }
                    }
                }
            }
            // urgency layer (mdp.py:2446-2447) for envs in their last 40 steps: nothing to do for most sub-groups
            const uint64_t urg_sub = (urgent_mask >> l0) & ((ne >= 64) ? ~0ull : ((1ull << ne) - 1ull));
            if (urg_sub != 0ull) {
                for (int q = lane; q < ne * cells_n; q += 64) {
                    const int le = q / cells_n;
                    const int c = q - le * cells_n;
                    if ((urg_sub >> le) & 1ull) {
                        T* env_img = imgT + (size_t)le * items_per_env * OC_NUM_LAYERS;
                        env_img[(size_t)c * OC_NUM_LAYERS + 25] = (T)1;
                        env_img[((size_t)cells_n + c) * OC_NUM_LAYERS + 25] = (T)1;
                    }
                }
            }
            wave_fence();
            // stream the image out: contiguous 16-byte stores (a ragged tail in dwords)
            const size_t total = env_bytes * ne;
            uint8_t* gdst = obs_k + env_bytes * (size_t)(wave_e0 + l0);
            const int n16 = (int)(total / 16);
            {   // four independent LDS reads in flight, then their four stores
                uint4* gd = reinterpret_cast<uint4*>(gdst);
                int i = lane;
                for (; i + 192 < n16; i += 256) {
                    const uint4 v0 = img[i], v1 = img[i + 64], v2 = img[i + 128], v3 = img[i + 192];
                    gd[i] = v0; gd[i + 64] = v1; gd[i + 128] = v2; gd[i + 192] = v3;
                }
                for (; i < n16; i += 64) gd[i] = img[i];
            }
            const int rem4 = (int)((total - (size_t)n16 * 16) / 4);
            if (lane < rem4)
                reinterpret_cast<uint32_t*>(gdst + (size_t)n16 * 16)[lane] =
                    reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(img) + (size_t)n16 * 16)[lane];
            wave_fence();
        }
        if (NW == 8) __syncthreads();  // the helpers are done reading before the next step writes
    }
    if (active) {
        store_env3<MAXP>(C, L, st, n, e, n_obj, s, cells);
        if (ep_returns) ep_returns[e] = ep;
    }
}
