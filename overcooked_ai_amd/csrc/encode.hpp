// encode.hpp — lossless_state_encoding: k_encode, k_encode_uniform
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
#pragma once

// ------------------------------------------------------------------------------------------
// k_encode: lossless_state_encoding (mdp.py:2385-2561) for both players of every env.
//
// Output per env: [2][W][H][26] values, i.e. 2*W*H "items" of 26 consecutive values each, item i of
// view v describing cell (x, y) = (i / H, i % H).  The encoding is >95 % zeros, so a workgroup that owns E
// consecutive envs (E * row ~ 40 KiB of LDS) (1) zero-fills an LDS image of its slice of the output with
// 16-byte stores, (2) scatters the few non-zero values — one task per (env, grid cell) for the terrain /
// object / urgency layers and one per (env, player) for the location / orientation / held-object layers —
// and (3) streams the image to HBM as contiguous 16-byte stores.  The output is the only real traffic:
// 2*W*H*26*sizeof(T) bytes per env against <= 144 bytes of state.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void enc_object_layers(T* item, uint32_t o, bool in_pot, uint32_t tk, uint32_t ct) {
    if (o & OC_O_SOUP) {
        const uint32_t n = (o >> 3) & 3u, nt = __popc(o & 7u), no = n - nt;
        if (in_pot && tk == 0u) { item[16] = (T)no; item[17] = (T)nt; }           // idle: *_in_pot (mdp.py:2490-2497)
        else {
            item[18] = (T)no; item[19] = (T)nt;                                     // mdp.py:2499-2525
            if (in_pot) {
                item[20] = (T)(ct - (tk - 1u));                                     // cook_time - _cooking_tick
                item[21] = (T)((tk - 1u) >= ct ? 1u : 0u);
            } else item[21] = (T)1;
        }
    } else if (o == OC_O_DISH) item[22] = (T)1;
    else if (o == OC_O_ONION) item[23] = (T)1;
    else if (o == OC_O_TOMATO) item[24] = (T)1;
}

// The same layers without branches: four stores whatever the object is.  Unused ones store 0 into layers 20 / 21,
// which no other object of the same cell can own (a cell holds at most one object; a player stands on a floor cell,
// which holds none).  A wavefront executes the union of its lanes' paths, so this is what the scatter loops call.
//   soup idle in a pot:  [16] = onions, [17] = tomatoes                       (mdp.py:2490-2497)
//   any other soup:      [18] = onions, [19] = tomatoes, [20] = time left (cooking in a pot), [21] = done (2499-2525)
//   dish / onion / tomato: [22] / [23] / [24] = 1                            (2527-2534)
template <typename T>
__device__ __forceinline__ void enc_object_writes(T* item, uint32_t o, bool in_pot, uint32_t tk, uint32_t ct) {
    const bool soup = (o & OC_O_SOUP) != 0u;
    const uint32_t n = (o >> 3) & 3u, nt = __popc(o & 7u), no = n - nt;
    const bool idle = soup & in_pot & (tk == 0u);
    const bool hot = soup & !idle;
    const uint32_t first = soup ? (idle ? 16u : 18u) : (o == OC_O_DISH ? 22u : o == OC_O_ONION ? 23u : 24u);
    item[first] = (T)(soup ? no : 1u);
    item[soup ? first + 1u : 20u] = (T)(soup ? nt : 0u);
    const uint32_t ticks = tk - 1u;
    item[20] = (T)((hot & in_pot) ? ct - ticks : 0u);
    item[21] = (T)(hot ? (in_pot ? (ticks >= ct ? 1u : 0u) : 1u) : 0u);
}

template <typename T, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_encode(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                  const uint16_t* __restrict__ layout_id,
                                                  const uint4* __restrict__ st, T* __restrict__ obs, int64_t n,
                                                  int W, int H, int n_planes, int envs_per_block, int horizon) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    const int cells = W * H;
    const int64_t e0 = (int64_t)blockIdx.x * envs_per_block;
    const int ne = (int)min((int64_t)envs_per_block, n - e0);
    // LDS carve: [envs_per_block][n_planes] uint4 state, then the output image
    uint4* s_state = reinterpret_cast<uint4*>(smem);
    const int state_bytes = envs_per_block * n_planes * 16;
    T* s_out = reinterpret_cast<T*>(smem + state_bytes);
    const int items_per_env = 2 * cells;
    const size_t env_bytes = (size_t)items_per_env * OC_NUM_LAYERS * sizeof(T);
    const size_t total = env_bytes * ne;  // multiple of 4; multiple of 16 unless this is a ragged tail block

    if (LAY_LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(g_layouts);
        for (int i = threadIdx.x; i < n_layouts * 16; i += BLOCK) s_lay[i] = src[i];
    }
    // issue the state loads first, zero-fill the image while they are in flight, then park them in LDS
    const int n_ld = ne * n_planes;
    uint4 ld0 = make_uint4(0, 0, 0, 0), ld1 = ld0;
    const int i0 = threadIdx.x, i1 = threadIdx.x + BLOCK;
    if (i0 < n_ld) ld0 = st[(int64_t)(i0 / ne) * n + e0 + (i0 % ne)];  // consecutive lanes: consecutive envs of a plane
    if (i1 < n_ld) ld1 = st[(int64_t)(i1 / ne) * n + e0 + (i1 % ne)];
    {
        uint4* img = reinterpret_cast<uint4*>(s_out);
        const size_t n16 = (total + 15) / 16;
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (size_t i = threadIdx.x; i < n16; i += BLOCK) img[i] = z;
    }
    if (i0 < n_ld) s_state[(i0 % ne) * n_planes + (i0 / ne)] = ld0;
    if (i1 < n_ld) s_state[(i1 % ne) * n_planes + (i1 / ne)] = ld1;
    for (int i = threadIdx.x + 2 * BLOCK; i < n_ld; i += BLOCK)
        s_state[(i % ne) * n_planes + (i / ne)] = st[(int64_t)(i / ne) * n + e0 + (i % ne)];
    __syncthreads();

    const uint32_t inv_w = 65536u / (uint32_t)W + 1u;  // y = c / W for c < 128 (exact: fractional parts are >= 1/W)
    const int tasks_per_env = cells + 2;
    for (int q = threadIdx.x; q < ne * tasks_per_env; q += BLOCK) {
        const int le = q / tasks_per_env;
        const int j = q - le * tasks_per_env;
        const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
        uint32_t lid = 0;
        if (layout_id != nullptr) lid = layout_id[e0 + le];
        const Lay L = LAY_LDS ? Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u}
                              : Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
        T* env_img = s_out + (size_t)le * items_per_env * OC_NUM_LAYERS;
        if (j < cells) {
            // terrain (mdp.py:2449-2465), urgency (2446-2447) and the object lying on this cell (2482-2534)
            const uint32_t c = (uint32_t)j;
            const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W;
            const uint32_t i = x * (uint32_t)H + y;
            const uint32_t tc = L.terrain(c), type = tc & 7u;
            const uint32_t o = se[16 + c];
            const uint32_t t = se[6] | ((uint32_t)se[7] << 8);
            const bool urgent = (horizon - (int)t) < 40;
            if (type != OC_T_FLOOR || urgent || o) {
                // layer of each terrain code: P(4)->10, X(1)->11, O(2)->12, T(3)->13, D(5)->14, S(6)->15
                const uint32_t layer = (0x0F0E0A0D0C0B00ull >> (8u * type)) & 0xFFu;
                uint32_t tk = 0, ct = 0;
                const bool in_pot = type == OC_T_POT;
                if (in_pot && o) { tk = se[8 + (tc >> 3)]; ct = L.cook_time(recipe_idx(o)); }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    T* item = env_img + ((size_t)v * cells + i) * OC_NUM_LAYERS;
                    if (type != OC_T_FLOOR) item[layer] = (T)1;
                    if (urgent) item[25] = (T)1;
                    if (o) enc_object_layers<T>(item, o, in_pot, tk, ct);
                }
            }
        } else {
            // player layers (mdp.py:2468-2479, ordering 2423-2434) and the held object (all_objects_list, 876-879)
            const int pl = j - cells;
            const uint32_t pos = se[3 * pl], ori = se[3 * pl + 1], held = se[3 * pl + 2];
            if (pos != 0xFFu) {
                const uint32_t y = (pos * inv_w) >> 16, x = pos - y * (uint32_t)W;
                const uint32_t i = x * (uint32_t)H + y;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    T* item = env_img + ((size_t)v * cells + i) * OC_NUM_LAYERS;
                    const int k = (pl == v) ? 0 : 1;  // the view's own player comes first
                    item[k] = (T)1;
                    item[2 + 4 * k + ori] = (T)1;
                    if (held) enc_object_layers<T>(item, held, false, 0u, 0u);
                }
            }
        }
    }
    __syncthreads();

    // stream the image out: contiguous, 16 B per lane per store
    uint8_t* gdst = reinterpret_cast<uint8_t*>(obs) + env_bytes * (size_t)e0;
    const uint8_t* ssrc = reinterpret_cast<const uint8_t*>(s_out);
    const size_t n16 = total / 16;
    // An output larger than the 256 MB MALL (f32 observations of 65 536 9x5 envs: 616 MB) streams past it — `sc1 nt`: 5.5 -> 6.0
    // TB/s; one that fits is rewritten in place inside it and must NOT (u8, 153 MB: 5.7 -> 4.1 TB/s with streaming stores)
    if (env_bytes * (size_t)n > ((size_t)320 << 20)) {
        for (size_t i = threadIdx.x; i < n16; i += BLOCK)
            stream_store16(reinterpret_cast<uint4*>(gdst) + i, reinterpret_cast<const uint4*>(ssrc)[i]);
    } else {
        for (size_t i = threadIdx.x; i < n16; i += BLOCK)
            reinterpret_cast<uint4*>(gdst)[i] = reinterpret_cast<const uint4*>(ssrc)[i];
    }
    const size_t rem4 = (total - n16 * 16) / 4;
    if (threadIdx.x < rem4)
        reinterpret_cast<uint32_t*>(gdst + n16 * 16)[threadIdx.x] =
            reinterpret_cast<const uint32_t*>(ssrc + n16 * 16)[threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// k_encode_uniform: k_encode specialised for a single layout shared by the whole batch (BASELINE configs[1-2]).
// The generic kernel above is instruction-issue bound, not HBM bound (SQ counters: ~630 instructions per
// wavefront per 4.7 KB of output, most of them the branchy scatter of the *static* terrain layers).  With one
// layout those layers are the same for every env, so persistent workgroups build them ONCE into an LDS template;
// per group of envs they copy template -> image (16-byte LDS moves), scatter only the dynamic values (one task
// per player and one per non-empty object dword; urgency only for envs in their last 40 steps) and stream the
// image out.  The template covers UNIT consecutive envs so that its size is a multiple of 16 bytes.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_encode_uniform(const OcLayout* __restrict__ g_layouts,
                                                          const uint4* __restrict__ st, T* __restrict__ obs,
                                                          int64_t n, int W, int H, int n_planes, int unit,
                                                          int units_per_group, int horizon) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint4 s_lay[16];
    const int cells = W * H;
    const int items_per_env = 2 * cells;
    const size_t env_bytes = (size_t)items_per_env * OC_NUM_LAYERS * sizeof(T);
    const size_t unit_bytes = env_bytes * unit;                 // multiple of 16 by construction
    const int unit_chunks = (int)(unit_bytes / 16);
    const int epg = unit * units_per_group;                     // envs per group
    // LDS carve: template (one unit), image (one group), state planes of the group
    uint4* s_tmpl = reinterpret_cast<uint4*>(smem);
    uint4* s_img = s_tmpl + unit_chunks;
    uint4* s_state = s_img + (size_t)unit_chunks * units_per_group;
    T* tmpl = reinterpret_cast<T*>(s_tmpl);
    T* img = reinterpret_cast<T*>(s_img);
    const uint32_t inv_w = 65536u / (uint32_t)W + 1u;

    if (threadIdx.x < 16) s_lay[threadIdx.x] = reinterpret_cast<const uint4*>(g_layouts)[threadIdx.x];
    for (int i = threadIdx.x; i < unit_chunks; i += BLOCK) s_tmpl[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const Lay L{reinterpret_cast<const uint8_t*>(s_lay)};
    // static terrain layers (mdp.py:2449-2465) of `unit` envs, both views
    for (int q = threadIdx.x; q < unit * cells; q += BLOCK) {
        const int u = q / cells;
        const uint32_t c = (uint32_t)(q - u * cells);
        const uint32_t type = L.terrain(c) & 7u;
        if (type != OC_T_FLOOR) {
            const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W, i = x * (uint32_t)H + y;
            const uint32_t layer = (0x0F0E0A0D0C0B00ull >> (8u * type)) & 0xFFu;  // P->10 X->11 O->12 T->13 D->14 S->15
            T* base = tmpl + (size_t)u * items_per_env * OC_NUM_LAYERS;
            base[((size_t)i) * OC_NUM_LAYERS + layer] = (T)1;
            base[((size_t)cells + i) * OC_NUM_LAYERS + layer] = (T)1;
        }
    }
    __syncthreads();

    const int64_t n_groups = (n + epg - 1) / epg;
    const int obj_dwords = (n_planes - 1) * 4;
    // which groups this workgroup encodes: with a grid that is a multiple of 8, XCD x (= workgroup id % 8) owns the x-th
    // contiguous eighth of the groups and its workgroups stride through it (xcd_block, common.hpp); else a plain grid stride
#ifdef OC_NO_XCD_REMAP
    const bool by_xcd = false;
#else
    const bool by_xcd = (gridDim.x & 7u) == 0u;
#endif
    const int64_t per_xcd = by_xcd ? (n_groups + 7) / 8 : n_groups;
    const int64_t g_first = by_xcd ? (int64_t)(blockIdx.x & 7u) * per_xcd : 0;
    const int64_t g_end = min(n_groups, g_first + per_xcd);
    const int64_t g_stride = by_xcd ? (int64_t)(gridDim.x >> 3) : (int64_t)gridDim.x;
    for (int64_t g = g_first + (by_xcd ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x); g < g_end; g += g_stride) {
        const int64_t e0 = g * epg;
        const int ne = (int)min((int64_t)epg, n - e0);
        // state planes of the group (issued first, parked after the template copy)
        const int n_ld = ne * n_planes;
        uint4 ld0 = make_uint4(0, 0, 0, 0);
        if ((int)threadIdx.x < n_ld) ld0 = st[(int64_t)(threadIdx.x / ne) * n + e0 + (threadIdx.x % ne)];
        for (int i = threadIdx.x; i < unit_chunks; i += BLOCK) {
            const uint4 v = s_tmpl[i];
            for (int u = 0; u < units_per_group; ++u) s_img[(size_t)u * unit_chunks + i] = v;
        }
        if ((int)threadIdx.x < n_ld) s_state[(threadIdx.x % ne) * n_planes + (threadIdx.x / ne)] = ld0;
        for (int i = threadIdx.x + BLOCK; i < n_ld; i += BLOCK)
            s_state[(i % ne) * n_planes + (i / ne)] = st[(int64_t)(i / ne) * n + e0 + (i % ne)];
        __syncthreads();

        // dynamic values.  A wavefront executes the union of its lanes' paths: players and objects run in separate
        // loops and an object's layers are written without branches (enc_object_writes).
        for (int t = threadIdx.x; t < 2 * ne; t += BLOCK) {  // players (mdp.py:2468-2479, ordering 2423-2434)
            const int le = t >> 1, pl = t & 1;
            const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
            const uint32_t pos = se[3 * pl], ori = se[3 * pl + 1], held = se[3 * pl + 2];
            if (pos != 0xFFu) {
                const uint32_t y = (pos * inv_w) >> 16, x = pos - y * (uint32_t)W, i = x * (uint32_t)H + y;
                T* own = img + ((size_t)le * items_per_env + (size_t)pl * cells + i) * OC_NUM_LAYERS;
                T* other = img + ((size_t)le * items_per_env + (size_t)(1 - pl) * cells + i) * OC_NUM_LAYERS;
                own[0] = (T)1; own[2 + ori] = (T)1;
                other[1] = (T)1; other[6 + ori] = (T)1;
                if (held) { enc_object_writes<T>(own, held, false, 0u, 0u); enc_object_writes<T>(other, held, false, 0u, 0u); }
            }
        }
        for (int q = threadIdx.x; q < ne * obj_dwords; q += BLOCK) {  // objects on the grid (mdp.py:2482-2534)
            const int le = q / obj_dwords;
            const int j = q - le * obj_dwords;
            const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
            uint32_t w = reinterpret_cast<const uint32_t*>(se + 16)[j];
            if (w != 0u) {
                T* env_img = img + (size_t)le * items_per_env * OC_NUM_LAYERS;
                while (w != 0u) {
                    const uint32_t b4 = (uint32_t)(__ffs((int)w) - 1) >> 3;  // lowest non-empty cell of the dword
                    const uint32_t o = (w >> (8u * b4)) & 0xFFu;
                    w &= ~(0xFFu << (8u * b4));
                    const uint32_t c = 4u * (uint32_t)j + b4;
                    const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W, i = x * (uint32_t)H + y;
                    const uint32_t tc = L.terrain(c);
                    const bool in_pot = (tc & 7u) == OC_T_POT;
                    const uint32_t tk = se[8 + (tc >> 3)];
                    const uint32_t ct = L.cook_time(recipe_idx(o) & 15u);
                    enc_object_writes<T>(env_img + (size_t)i * OC_NUM_LAYERS, o, in_pot, tk, ct);
                    enc_object_writes<T>(env_img + ((size_t)cells + i) * OC_NUM_LAYERS, o, in_pot, tk, ct);
                }
            }
        }
        // urgency layer (mdp.py:2446-2447) for envs in their last 40 steps
        for (int q = threadIdx.x; q < ne * cells; q += BLOCK) {
            const int le = q / cells;
            const int c = q - le * cells;
            const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
            const uint32_t t = se[6] | ((uint32_t)se[7] << 8);
            if ((horizon - (int)t) < 40) {
                T* env_img = img + (size_t)le * items_per_env * OC_NUM_LAYERS;
                env_img[(size_t)c * OC_NUM_LAYERS + 25] = (T)1;
                env_img[((size_t)cells + c) * OC_NUM_LAYERS + 25] = (T)1;
            }
        }
        __syncthreads();

        // stream the image out: contiguous 16-byte stores (ragged tails in dwords)
        const size_t total = env_bytes * ne;
        uint8_t* gdst = reinterpret_cast<uint8_t*>(obs) + env_bytes * (size_t)e0;
        const size_t n16 = total / 16;
        for (size_t i = threadIdx.x; i < n16; i += BLOCK) reinterpret_cast<uint4*>(gdst)[i] = s_img[i];
        const size_t rem4 = (total - n16 * 16) / 4;
        if (threadIdx.x < rem4)
            reinterpret_cast<uint32_t*>(gdst + n16 * 16)[threadIdx.x] =
                reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s_img) + n16 * 16)[threadIdx.x];
        __syncthreads();
    }
}


