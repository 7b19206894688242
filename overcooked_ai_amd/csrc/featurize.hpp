// featurize.hpp — featurize_state: k_featurize
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
#pragma once

// ------------------------------------------------------------------------------------------
// k_featurize: featurize_state (mdp.py:2579-2898), the hand-crafted feature vector used by the behaviour-cloning
// agents.  Two lanes per env (lane parity = player).  Each lane walks the grid once; for every feature cell it
// reads COST[state][cell] (fewest actions from the player's (cell, orientation) to a goal of that feature,
// precomputed on the host from the reference's MotionPlanner semantics, overcooked_ai_amd/planner.py) and keeps the
// arg-min per category.  min_cost_to_feature (planners.py:391-423) breaks ties by list order — dispensers before
// counter objects, row-major inside each group — which is the lexicographic minimum of (cost, group, cell).
// The 2 x (2*(num_pots*10+26)+4) floats of an env are assembled in an LDS image and streamed out coalesced.
// ------------------------------------------------------------------------------------------
constexpr int FEAT_ENVS = BLOCK / 2;

__device__ __forceinline__ uint32_t feat_key(uint32_t cost, uint32_t group, uint32_t cell) {
    return (cost << 9) | (group << 8) | cell;  // cost < 255, cell < 128
}

template <bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_featurize(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                     const uint16_t* __restrict__ layout_id,
                                                     const uint8_t* __restrict__ plan_blob,
                                                     const uint32_t* __restrict__ plan_off,
                                                     const uint4* __restrict__ st, float* __restrict__ out, int64_t n,
                                                     int W, int H, int n_planes, int num_pots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    const int per = num_pots * 10 + 26, total = 2 * per + 4;  // floats per player block / per (env, player) row
    // every feature is a small integer (deltas, counts, flags, cook time left < 255): the LDS image holds int16 and
    // the copy-out converts.  Row stride total + 2 shorts = an odd number of dwords: lanes writing column k of
    // consecutive rows hit 32 different banks.
    const int rs = total + 2;
    uint4* s_state = reinterpret_cast<uint4*>(smem);           // [FEAT_ENVS][n_planes]
    int16_t* s_img = reinterpret_cast<int16_t*>(smem + (size_t)FEAT_ENVS * n_planes * 16);  // [FEAT_ENVS][2][rs]
    const uint32_t p = threadIdx.x & 1u, el = threadIdx.x >> 1;
    const int64_t e0 = (int64_t)blockIdx.x * FEAT_ENVS;
    const int ne = (int)min((int64_t)FEAT_ENVS, n - e0);
    const int64_t e = e0 + el;
    const bool active = (int)el < ne;
    for (int i = threadIdx.x; i < ne * n_planes; i += BLOCK)
        s_state[(i % ne) * n_planes + (i / ne)] = st[(int64_t)(i / ne) * n + e0 + (i % ne)];
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (active) {
        const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + el * n_planes);
        const uint32_t lid = layout_id ? layout_id[e] : 0u;
        const uint8_t* plan = plan_blob + plan_off[lid];
        const uint32_t pos = se[3 * p], ori = se[3 * p + 1], held = se[3 * p + 2];
        const uint32_t opos = se[3 * (1 - p)];
        const uint32_t inv_w = 65536u / (uint32_t)W + 1u;
        const uint32_t py = (pos * inv_w) >> 16, px = pos - py * (uint32_t)W;
        const uint32_t state = (uint32_t)plan[pos] * 4u + ori;  // (free cell, orientation)
        const uint8_t* cost_row = plan + 128 + state * (uint32_t)(n_planes - 1) * 16u;
        // What depends on the terrain alone was found on the host (planner.walk_records): the closest dispenser of each
        // kind, the closest serving cell, the four closest pots, and the goal counters in the order the reference's
        // arg-min would prefer them.  (Up to round 3 every lane walked the whole grid for this: 7.4 of the kernel's 18.7 us.)
        const uint8_t* wsec = plan_blob + plan_off[n_layouts + lid];
        const uint8_t* rec = wsec + 16 + state * *reinterpret_cast<const uint32_t*>(wsec);
        const uint4 sb = *reinterpret_cast<const uint4*>(rec), pk = *reinterpret_cast<const uint4*>(rec + 16);
        // arg-min keys: 0 onion, 1 tomato, 2 dish, 3 counter soup, 4 serving, 5 empty counter; the four best pots
        uint32_t best[6] = {sb.x, sb.y, sb.z, ~0u, sb.w, ~0u};
        const uint32_t pot1 = pk.x, pot2 = pk.y, pot3 = pk.z, pot4 = pk.w;
        const uint32_t n_goal = rec[32];
        if (n_goal != 0u) {  // counters are motion goals (not the reference's default NO_COUNTERS_PARAMS)
            for (uint32_t i = 0; i < n_goal; ++i) {  // the closest empty counter: the first of the sorted list without an object
                const uint32_t c = rec[33 + i];
                if (se[16 + c] == 0u) { best[5] = feat_key(cost_row[c], 0, c); break; }
            }
            const uint32_t obj_dwords = (uint32_t)(n_planes - 1) * 4u;
            for (uint32_t j = 0; j < obj_dwords; ++j) {  // what lies on the counters competes with the dispensers (group 1)
                uint32_t w = reinterpret_cast<const uint32_t*>(se + 16)[j];
                while (w != 0u) {
                    const uint32_t b4 = (uint32_t)(__ffs((int)w) - 1) >> 3;
                    const uint32_t o = (w >> (8u * b4)) & 0xFFu, c = 4u * j + b4;
                    w &= ~(0xFFu << (8u * b4));
                    if ((L.terrain(c) & 7u) != OC_T_COUNTER) continue;  // (a pot's soup is no counter object)
                    const uint32_t cost = cost_row[c];
                    if (cost == 255u) continue;
                    if (o == OC_O_ONION) best[0] = min(best[0], feat_key(cost, 1, c));
                    else if (o == OC_O_TOMATO) best[1] = min(best[1], feat_key(cost, 1, c));
                    else if (o == OC_O_DISH) best[2] = min(best[2], feat_key(cost, 1, c));
                    else best[3] = min(best[3], feat_key(cost, 0, c));
                }
            }
        }
        int16_t* own = s_img + ((size_t)el * 2 + p) * rs;              // this player's row: own block first
        int16_t* oth = s_img + ((size_t)el * 2 + (1 - p)) * rs + per;  // the other row carries it second
        int k = 0;
        auto put = [&](int v) { own[k] = (int16_t)v; oth[k] = (int16_t)v; ++k; };
        for (uint32_t d = 0; d < 4; ++d) put(ori == d ? 1 : 0);
        // IDX_TO_OBJ = [onion, soup, dish, tomato] (mdp.py:2733)
        put(held == OC_O_ONION ? 1 : 0); put((held & OC_O_SOUP) ? 1 : 0);
        put(held == OC_O_DISH ? 1 : 0); put(held == OC_O_TOMATO ? 1 : 0);
        const bool held_is[6] = {held == OC_O_ONION, held == OC_O_TOMATO, held == OC_O_DISH, (held & OC_O_SOUP) != 0u,
                                 false, false};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int dx = 0, dy = 0;
            uint32_t soup = 0;
            if (held_is[q]) { soup = held; }  // a held object of this kind: deltas (0, 0) (mdp.py:2629-2632)
            else if (best[q] != ~0u) {
                const uint32_t c = best[q] & 0x7Fu;
                const uint32_t cy = (c * inv_w) >> 16, cx = c - cy * (uint32_t)W;
                dx = (int)cx - (int)px; dy = (int)cy - (int)py;
                soup = se[16 + c];
            }
            put(dx); put(dy);
            if (q == 3) {  // ingredient counts of the closest (or held) soup
                const uint32_t nn = (soup & OC_O_SOUP) ? ((soup >> 3) & 3u) : 0u;
                const uint32_t nt = (soup & OC_O_SOUP) ? __popc(soup & 7u) : 0u;
                put((int)(nn - nt)); put((int)nt);
            }
        }
        const uint32_t potk[4] = {pot1, pot2, pot3, pot4};
        for (int j = 0; j < num_pots; ++j) {
            const uint32_t key = j < 4 ? potk[j] : ~0u;
            if (key == ~0u) { for (int z = 0; z < 10; ++z) put(0); continue; }
            const uint32_t c = key & 0x7Fu;
            const uint32_t cy = (c * inv_w) >> 16, cx = c - cy * (uint32_t)W;
            const uint32_t o = se[16 + c], tk = se[8 + (L.terrain(c) >> 3)];
            const uint32_t nn = (o >> 3) & 3u, nt = __popc(o & 7u);
            const uint32_t ct = L.cook_time((nn - nt) + 4u * nt);
            const bool empty = o == 0u, idle = tk == 0u;
            const bool ready = !empty && !idle && (tk - 1u) >= ct, cooking = !empty && !idle && !ready;
            const bool full = cooking || ready || (!empty && nn == 3u);
            const uint32_t remaining = (empty || idle || ready) ? 0u : ct - (tk - 1u);
            put(1); put(empty ? 1 : 0); put(full ? 1 : 0); put(cooking ? 1 : 0); put(ready ? 1 : 0);
            put(empty ? 0 : (int)(nn - nt)); put(empty ? 0 : (int)nt); put((int)remaining);
            put((int)cx - (int)px); put((int)cy - (int)py);
        }
        for (uint32_t d = 0; d < 4; ++d) {  // walls (mdp.py:2831-2838)
            const uint32_t c = pos + (uint32_t)(d == 0 ? -W : d == 1 ? W : d == 2 ? 1 : -1);
            put((L.terrain(c) & 7u) == OC_T_FLOOR ? 0 : 1);
        }
        const uint32_t oy = (opos * inv_w) >> 16, ox = opos - oy * (uint32_t)W;
        int16_t* row = s_img + ((size_t)el * 2 + p) * rs;
        row[2 * per + 0] = (int16_t)((int)ox - (int)px);  // other player's position relative to this one
        row[2 * per + 1] = (int16_t)((int)oy - (int)py);
        row[2 * per + 2] = (int16_t)px;
        row[2 * per + 3] = (int16_t)py;
    }
    __syncthreads();
    // rows are contiguous in the output: stream them out as 16-byte stores (total is a multiple of 4)
    const uint32_t q_per_row = (uint32_t)total / 4u, n_q = (uint32_t)ne * 2u * q_per_row;
    const uint32_t magic = 0xFFFFFFFFu / q_per_row + 1u;  // i / q_per_row == mulhi(i, magic) for i < 2^16
    float4* gdst = reinterpret_cast<float4*>(out + (size_t)e0 * 2 * total);
    for (uint32_t i = threadIdx.x; i < n_q; i += BLOCK) {
        const uint32_t row = __umulhi(i, magic), col = i - row * q_per_row;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(s_img + (size_t)row * rs + 4u * col);
        const uint32_t w0 = src[0], w1 = src[1];
        gdst[i] = make_float4((float)(int16_t)(w0 & 0xFFFFu), (float)((int32_t)w0 >> 16),
                              (float)(int16_t)(w1 & 0xFFFFu), (float)((int32_t)w1 >> 16));
    }
}
