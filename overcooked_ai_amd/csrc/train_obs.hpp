// train_obs.hpp — the batched step of the RLlib training environment WITH its observation, one kernel: k_train_step_obs
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace after shaping.hpp (k_train_step1), encode.hpp
// (enc_object_writes), rollout_encode.hpp (wave_fence) and potential.hpp (potential2_core).
#pragma once

// ------------------------------------------------------------------------------------------
// k_train_step_obs: OvercookedMultiAgent.step (human_aware_rl/rllib/rllib.py:293-342) for a batch with ONE two-player layout —
// the transition (get_state_transition, mdp.py:1375), phi(s') and the shaped rewards, the restart of finished envs, and
// lossless_state_encoding (mdp.py:2385-2561) of the states the next step starts from — in one launch.  As two kernels
// (k_train_step1, then k_encode_uniform) the step costs 9 + 16.6 us on 65 536 cramped_room envs: two launches, the state
// written and read back, and two latency chains end to end — the step's (loads, transition, ~3.8 us of float64 potential on
// one wavefront per SIMD, with HBM idle) and the encoder's (states, first image) — before the 68 MB of observations move.
// Here a workgroup of EIGHT wavefronts owns 256 envs:
//   * wavefronts 0..3 (owners, lane = env) run k_train_step1's transition on the wire format, restart finished envs, store the
//     state, and leave in LDS what the others need: the new object planes (their own rows), the new header, a record of s'
//     before any restart (players, pots, flags) and the reward quad;
//   * after ONE workgroup barrier, wavefronts 4..7 (helpers, same lane = env) compute phi(s') and the shaped rewards from
//     those records — the float64 arithmetic that had HBM waiting — while the owners already encode and stream;
//   * the observation of a wavefront's 64 envs is produced in sub-groups of G envs through a PRIVATE LDS image per wavefront
//     (template copy, scatter of the dynamic values, contiguous 16-byte stores — k_encode_uniform's steps, with wave-level
//     fences only); owner w and helper w claim the sub-groups of owner w's envs from one LDS counter, so the helper joins
//     as soon as its potentials are out.
// Same outputs, bit for bit, as k_train_step1 + k_encode_uniform (tests/test_gpu_parity.py compares the two).
// ------------------------------------------------------------------------------------------
template <int MAXP>
__device__ __forceinline__ uint4 one_header(const LayC& C, const Env3<MAXP>& s) {
    uint4 ho;
    ho.x = s.pos0 | (s.or0 << 8) | (s.held0 << 16) | (s.pos1 << 24);
    ho.y = s.or1 | (s.held1 << 8) | (min(s.t, 0xFFFFu) << 16);  // the wire format's u16 timestep saturates
    ho.z = 0; ho.w = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k)
        if ((uint32_t)k < C.n_pots) ho.z |= s.tk[k] << (8 * (k & 3));
    return ho;
}
// what phi needs of a state with at most two pots: players, the pots' soup codes and tick bytes (+ the step's flag bits)
template <int MAXP>
__device__ __forceinline__ uint4 phi_record(const Env3<MAXP>& s, uint32_t fl, uint32_t drawn) {
    uint4 r;
    r.x = s.pos0 | (s.or0 << 8) | (s.held0 << 16) | (s.pos1 << 24);
    r.y = s.or1 | (s.held1 << 8) | (s.ps[0] << 16) | ((MAXP > 1 ? s.ps[MAXP - 1] : 0u) << 24);
    r.z = s.tk[0] | ((MAXP > 1 ? s.tk[MAXP - 1] : 0u) << 8) | (fl << 16) | (drawn << 24);
    r.w = 0;
    return r;
}

#ifdef OC_AMD_TUNING
// tuning builds: where a wavefront's time goes, in 10 ns ticks — [workgroup][wavefront][8]: kernel start -> second barrier passed,
// barrier -> first claim (the helpers' potentials), then sums over its sub-groups: template copy, players, objects (+ urgency), stream;
// [6] sub-groups taken, [7] kernel start -> this wavefront done
__device__ uint32_t g_obs_dbg[4096 * 8 * 8];
#define OBS_T(var) const uint64_t var = mb_now()
#else
#define OBS_T(var)
#endif

// NWV wavefronts per workgroup: 0..3 owners, 4..7 helpers (the potentials), 8.. (NWV = 16, round 6) ENCODERS that own nothing and
// compute nothing but sub-groups.  tools/obs_phases.py (tuning build, stamps per phase) showed what the u8 observation of a small
// grid costs: not bytes — 3.5 us until the second barrier, then per 14-env sub-group ~1 us of template copy, 0.4 players, 1.6
// objects, 0.7 stream on a wavefront that shares its SIMD with ONE other (an instruction every ~8 clocks); a SIMD issues from up
// to ~5 wavefronts at that latency, so more wavefronts per CU with smaller private images, not fewer instructions, is the lever.
template <int MAXP, typename T, int NWV>
__global__ __launch_bounds__(NWV * 64) void k_train_step_obs(
    const OcLayout* __restrict__ g_layouts, uint4* st, const uint8_t* __restrict__ actions, float4* __restrict__ rewards,
    uint8_t* __restrict__ flags, float4* ep_returns, float4* __restrict__ ep_out, const uint8_t* __restrict__ plan_blob,
    const uint32_t* __restrict__ plan_off, const uint8_t* __restrict__ phi_tables, double* __restrict__ phi_next,
    double* __restrict__ phi_cur, const double* __restrict__ phi_start, double factor, double* __restrict__ shaped,
    uint8_t* __restrict__ done, uint8_t* __restrict__ obs_bytes, int64_t n, int W, int H, int n_obj, int horizon, int unit,
    int group_envs, StartArgs sa) {
#pragma clang fp contract(off)
    OBS_T(tm_start);
    extern __shared__ __attribute__((aligned(16))) uint4 s_dyn[];  // rows | template | header | records | rewards | images
    __shared__ uint4 s_lay[16];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    __shared__ uint16_t s_ioff[64];   // cell -> offset of its item in a view, in values: 26 * (x * H + y)
    __shared__ uint32_t s_next[4];    // per owner wavefront: the next sub-group of its envs nobody has taken yet
    const int cells_n = W * H;
    const int items_per_env = 2 * cells_n;
    const size_t env_bytes = (size_t)items_per_env * OC_NUM_LAYERS * sizeof(T);
    const int unit_chunks = (int)(env_bytes * unit / 16);      // the template: `unit` envs, a multiple of 16 bytes
    const int img_chunks = unit_chunks * (group_envs / unit);  // one wavefront's image: group_envs envs
    uint4* s_rows = s_dyn;                                     // [n_obj][BLOCK]: object planes, one 16-byte row per env
    uint4* s_tmpl = s_rows + (size_t)n_obj * BLOCK;
    uint4* s_hdr = s_tmpl + unit_chunks;                       // [BLOCK] header of the state the next step starts from
    uint4* s_pre = s_hdr + BLOCK;                              // [BLOCK] phi_record of s' before any restart
    uint4* s_post = s_pre + BLOCK;                             // [BLOCK] phi_record of a DRAWN start state
    float4* s_rw = reinterpret_cast<float4*>(s_post + BLOCK);  // [BLOCK] the step's reward quad
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* img = reinterpret_cast<uint4*>(s_rw + BLOCK) + (size_t)wave * img_chunks;
    T* imgT = reinterpret_cast<T*>(img);
    T* tmpl = reinterpret_cast<T*>(s_tmpl);
    const bool owner = threadIdx.x < BLOCK;
    const int ow = wave & 3;
    const uint32_t tid = threadIdx.x & (BLOCK - 1);
    const uint32_t blk = xcd_block();  // (common.hpp: each XCD owns a contiguous eighth of the envs and of the observations)
    const int64_t e = (int64_t)blk * BLOCK + tid;
    const bool active = e < n;
    const int64_t el = active ? e : n - 1;
    // ---- everything the step reads, requested before the first wait (owners); the helpers ask for phi(s)
    OneIn in;
    double phi_before = 0.0;
    if (owner) in = one_load(st, actions, ep_returns, n, el, n_obj);
    else if (wave < 8 && phi_tables) phi_before = phi_cur[el];
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += NWV * 64) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    if (threadIdx.x < 16) s_lay[threadIdx.x] = reinterpret_cast<const uint4*>(g_layouts)[threadIdx.x];
    if (threadIdx.x < 4) s_next[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < unit_chunks; i += NWV * 64) s_tmpl[i] = make_uint4(0, 0, 0, 0);
    const uint32_t inv_w = 65536u / (uint32_t)W + 1u;
    if (threadIdx.x < 64) {
        const uint32_t c = threadIdx.x, y = (c * inv_w) >> 16, x = c - y * (uint32_t)W;
        s_ioff[c] = (uint16_t)((x * (uint32_t)H + y) * OC_NUM_LAYERS);  // (cells beyond the grid are never looked up)
    }
    __syncthreads();
    const Lay L{reinterpret_cast<const uint8_t*>(s_lay)};
    // static terrain layers (mdp.py:2449-2465) of `unit` envs, both views — read by nobody before the second barrier
    for (int q = threadIdx.x; q < unit * cells_n; q += NWV * 64) {
        const int u = q / cells_n;
        const uint32_t c = (uint32_t)(q - u * cells_n);
        const uint32_t type = L.terrain(c) & 7u;
        if (type != OC_T_FLOOR) {
            const uint32_t layer = (0x0F0E0A0D0C0B00ull >> (8u * type)) & 0xFFu;  // P->10 X->11 O->12 T->13 D->14 S->15
            T* base = tmpl + (size_t)u * items_per_env * OC_NUM_LAYERS + s_ioff[c];
            base[layer] = (T)1;
            base[(size_t)cells_n * OC_NUM_LAYERS + layer] = (T)1;
        }
    }
    const LayC C = load_consts<true>(L);
    if (owner && active) {
        // ---- the transition on the wire format (k_train_step1), the restart, the state written once
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            if (p < n_obj) s_rows[p * BLOCK + tid] = in.v[p];
        uint8_t* row = reinterpret_cast<uint8_t*>(s_rows + tid);
        const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
        One<MAXP> q;
        one_decode<MAXP>(C, L, in.h, row, q);
        Env3<MAXP>& s = q.s;
        const uint32_t a0 = in.a01 & 0xFFu, a1 = in.a01 >> 8;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f), ep = in.ep;
        uint32_t fl = 0;
        if (a0 > 5u || a1 > 5u) {
            fl = OC_F_BAD_ACTION;  // the env stays untouched (mdp.py:1394-1398 raises)
        } else {
            one_transition<MAXP>(C, L, lut, make_delta4(W), a0, a1, in.v, n_obj, row, q, r);
            ep.x += r.x; ep.y += r.y; ep.z += r.z; ep.w += r.w;
            if ((int)s.t >= horizon) fl |= OC_F_DONE;
        }
        const bool is_done = (fl & OC_F_DONE) != 0u;
        s_pre[tid] = phi_record<MAXP>(s, fl, (is_done && sa.enabled) ? 1u : 0u);
        s_rw[tid] = r;
        done[e] = is_done ? 1 : 0;
        if (ep_out) ep_out[e] = ep;
        if (!phi_tables) {  // the shaped rewards of the step itself (no potential: nothing for the helpers to compute)
            const double sparse = (double)r.x + (double)r.y;
            reinterpret_cast<double2*>(shaped)[e] = make_double2(sparse + factor * (double)r.z, sparse + factor * (double)r.w);
        }
        if (is_done) {  // the next episode: the standard start state, or one drawn from the batch's start_state_fn
            one_restart<MAXP>(C, L, sa, (uint64_t)(sa.env_offset + e), s);
            if (sa.enabled) s_post[tid] = phi_record<MAXP>(s, 0u, 0u);
            ep = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        one_store<MAXP>(C, L, st, n, e, n_obj, q, is_done, row);  // header + every plane, the planes left in this lane's rows
        s_hdr[tid] = one_header<MAXP>(C, s);
        rewards[e] = r;
        flags[e] = (uint8_t)fl;
        if (ep_returns) ep_returns[e] = ep;
    }
    __syncthreads();  // the only barrier behind the staging one: rows, headers, records and the template are in LDS
    OBS_T(tm_bar);
    if (!owner && wave < 8 && active && phi_tables) {
        // ---- phi(s'), the shaped rewards, phi(s) of the next step (k_train_step1's arithmetic, from the records)
        const Phi Tb{phi_tables};
        const uint8_t* plan = plan_blob + plan_off[0];
        auto phi_of = [&](const uint4 rec) {
            return potential2_core(L, Tb, plan, (uint32_t)(W * H), 2u, rec.x & 0xFFu, (rec.x >> 8) & 0xFFu, (rec.x >> 16) & 0xFFu,
                                   rec.x >> 24, rec.y & 0xFFu, (rec.y >> 8) & 0xFFu, (rec.y >> 16) & 0xFFu, rec.y >> 24,
                                   rec.z & 0xFFu, (rec.z >> 8) & 0xFFu);
        };
        const uint4 pre = s_pre[tid];
        const float4 r = s_rw[tid];
        const bool is_done = ((pre.z >> 16) & OC_F_DONE) != 0u;
        const double pn = phi_of(pre);
        const double sparse = (double)r.x + (double)r.y;
        const double d = pn - phi_before;
        phi_next[e] = pn;
        double pc = is_done ? phi_start[0] : pn;
        if ((pre.z >> 24) != 0u) pc = phi_of(s_post[tid]);  // a drawn start state: phi(s) of the next step is ITS potential
        phi_cur[e] = pc;
        reinterpret_cast<double2*>(shaped)[e] = make_double2(sparse + factor * d, sparse + factor * d);
    }
    // ---- lossless_state_encoding of the owner wavefront's 64 envs, sub-group by sub-group, owner and helper taking turns
    const int64_t wave_e0 = (int64_t)blk * BLOCK + (int64_t)ow * 64;
    const int n_wave = (int)max((int64_t)0, min((int64_t)64, n - wave_e0));
    const int n_groups = (n_wave + group_envs - 1) / group_envs;
    const int obj_dwords = n_obj * 4;
    const uint4* whdr = s_hdr + ow * 64;
    const uint32_t tmpl_magic = 0xFFFFFFFFu / (uint32_t)unit_chunks + 1u;  // i / unit_chunks == mulhi(i, magic) for i < 2^16
    OBS_T(tm_loop);
#ifdef OC_AMD_TUNING
    uint32_t acc_copy = 0, acc_pl = 0, acc_obj = 0, acc_str = 0, acc_n = 0;
#endif
    for (;;) {
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(&s_next[ow], 1u);
        g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        if ((int)g >= n_groups) break;
        const int l0 = (int)g * group_envs;
        const int ne = min(group_envs, n_wave - l0);
        const int n_units = (ne + unit - 1) / unit;
        OBS_T(t0);
        // the image = n_units copies of the template back to back: chunk i <- template chunk i mod unit_chunks (every lane busy
        // in every round; a loop over the template's chunks leaves 63 lanes idle in its last round when unit_chunks = 65)
        {
            const int n_img = n_units * unit_chunks;
            for (int i = lane; i < n_img; i += 64) {
                const uint32_t u = __umulhi((uint32_t)i, tmpl_magic);  // i / unit_chunks (i < 2^16)
                img[i] = s_tmpl[i - (int)u * unit_chunks];
            }
        }
        wave_fence();
        OBS_T(t1);
        // players (mdp.py:2468-2479, ordering 2423-2434) and what they hold: lane = (env, player)
        bool urgent = false;
        for (int t = lane; t < 2 * ne; t += 64) {
            const int le = t >> 1, pl = t & 1;
            const uint4 hw = whdr[l0 + le];
            const uint32_t pos = pl == 0 ? (hw.x & 0xFFu) : (hw.x >> 24);
            const uint32_t ori = pl == 0 ? ((hw.x >> 8) & 0xFFu) : (hw.y & 0xFFu);
            const uint32_t held = pl == 0 ? ((hw.x >> 16) & 0xFFu) : ((hw.y >> 8) & 0xFFu);
            urgent |= (horizon - (int)(hw.y >> 16)) < 40;
            if (pos != 0xFFu) {
                const uint32_t io = s_ioff[pos & 63u];
                T* own = imgT + ((uint32_t)le * (uint32_t)items_per_env + (uint32_t)pl * (uint32_t)cells_n) * OC_NUM_LAYERS + io;
                T* other = imgT + ((uint32_t)le * (uint32_t)items_per_env + (uint32_t)(1 - pl) * (uint32_t)cells_n) * OC_NUM_LAYERS + io;
                own[0] = (T)1; own[2 + ori] = (T)1;
                other[1] = (T)1; other[6 + ori] = (T)1;
                if (held) { enc_object_writes<T>(own, held, false, 0u, 0u); enc_object_writes<T>(other, held, false, 0u, 0u); }
            }
        }
        OBS_T(t2);
        // objects on the grid (mdp.py:2482-2534): lane = (env, object dword)
        for (int q = lane; q < ne * obj_dwords; q += 64) {
            const int le = q / obj_dwords, j = q - le * obj_dwords;
            const int l = ow * 64 + l0 + le;
            uint32_t w = reinterpret_cast<const uint32_t*>(s_rows + (j >> 2) * BLOCK + l)[j & 3];
            if (w != 0u) {
                const uint32_t tkw = s_hdr[l].z;
                T* env_img = imgT + (size_t)le * items_per_env * OC_NUM_LAYERS;
                while (w != 0u) {
                    const uint32_t b4 = (uint32_t)(__ffs((int)w) - 1) >> 3;  // lowest non-empty cell of the dword
                    const uint32_t o = (w >> (8u * b4)) & 0xFFu;
                    w &= ~(0xFFu << (8u * b4));
                    const uint32_t c = 4u * (uint32_t)j + b4;
                    const uint32_t tc = L.terrain(c);
                    const bool in_pot = (tc & 7u) == OC_T_POT;
                    const uint32_t tk = (tkw >> (8u * ((tc >> 3) & 3u))) & 0xFFu;
                    const uint32_t ct = L.cook_time(recipe_idx(o) & 15u);
                    T* item = env_img + s_ioff[c & 63u];
                    enc_object_writes<T>(item, o, in_pot, tk, ct);
                    enc_object_writes<T>(item + (uint32_t)cells_n * OC_NUM_LAYERS, o, in_pot, tk, ct);
                }
            }
        }
        // urgency layer (mdp.py:2446-2447) for envs in their last 40 steps: nothing to do for most sub-groups
        if (__ballot(urgent) != 0ull) {
            for (int q = lane; q < ne * cells_n; q += 64) {
                const int le = q / cells_n, c = q - le * cells_n;
                if ((horizon - (int)(whdr[l0 + le].y >> 16)) < 40) {
                    T* env_img = imgT + (size_t)le * items_per_env * OC_NUM_LAYERS;
                    env_img[(size_t)c * OC_NUM_LAYERS + 25] = (T)1;
                    env_img[((size_t)cells_n + c) * OC_NUM_LAYERS + 25] = (T)1;
                }
            }
        }
        wave_fence();
        OBS_T(t3);
        // stream the image out: contiguous 16-byte stores (a ragged tail in dwords)
        const size_t total = env_bytes * ne;
        uint8_t* gdst = obs_bytes + env_bytes * (size_t)(wave_e0 + l0);
        const int n16 = (int)(total / 16);
        {
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
#ifdef OC_AMD_TUNING
        {
            const uint64_t t4 = wall_clock64();  // (not mb_now: the stores just issued are not waited for)
            acc_copy += (uint32_t)(t1 - t0); acc_pl += (uint32_t)(t2 - t1); acc_obj += (uint32_t)(t3 - t2); acc_str += (uint32_t)(t4 - t3); ++acc_n;
        }
#endif
    }
#ifdef OC_AMD_TUNING
    if (lane == 0 && blockIdx.x < 4096) {
        uint32_t* d = g_obs_dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = (uint32_t)(tm_bar - tm_start); d[1] = (uint32_t)(tm_loop - tm_bar); d[2] = acc_copy; d[3] = acc_pl; d[4] = acc_obj; d[5] = acc_str;
        d[6] = acc_n; d[7] = (uint32_t)(wall_clock64() - tm_start);
    }
#endif
}
