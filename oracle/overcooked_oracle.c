/*
 * overcooked_oracle.c — CPU restatement of the Overcooked hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it.  The shipped package (overcooked_ai_amd) never imports oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this code against fixtures under
 * tests/golden/ that were produced by running the real reference (imported from /root/reference
 * by oracle/gen_golden.py): the reference's own 1500-step golden trajectory
 * (data/testing/test_mdp_dynamics/expected.json), its one-transition fixture, start-state fixture,
 * ~10^5 randomized transitions over 11 layout configurations (incl. old_dynamics, tomato/bonus
 * layouts), the K1..K11 micro cases of SURVEY.md §8c, lossless encodings of visited states, whole
 * OvercookedEnv episodes with game_stats, featurize_state (incl. the reference's own golden pickle
 * test_state_featurization/expected_2.pickle) and potential_function (6 400 values, bit-identical float64).
 * oracle_reset_random has no reference counterpart to be pinned against draw for draw (the reference draws
 * from numpy's global generator): it restates the stream documented at oc_reset_random and its statistics are
 * checked against the distribution of get_random_start_state_fn.
 *
 * It deliberately mirrors the *structure* of the reference (objects keyed by position, players
 * processed in index order, the helper predicates of the reference) rather than the structure of
 * the HIP kernels, so that kernel-vs-oracle agreement is a real cross-check.  All file:line
 * citations are into /root/reference/src/overcooked_ai_py/mdp/overcooked_mdp.py ("mdp.py") unless
 * stated otherwise.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_CELLS 128
#define MAX_POTS 8
#define MAX_RECIPES 9 /* multisets of 1..3 items over {onion, tomato}: 2+3+4 */
#define NUM_LAYERS 26

enum { NAME_NONE = 0, NAME_ONION = 1, NAME_TOMATO = 2, NAME_DISH = 3, NAME_SOUP = 4 };
enum { A_NORTH = 0, A_SOUTH, A_EAST, A_WEST, A_STAY, A_INTERACT };

/* Direction.INDEX_TO_DIRECTION, actions.py:12-16 ; STAY = (0,0), actions.py:47 */
static const int DIR_DX[5] = {0, 0, 1, -1, 0};
static const int DIR_DY[5] = {-1, 1, 0, 0, 0};

/* ------------------------------------------------------------------------------------------
 * Layout / recipe configuration, in the reference's own terms (the kwargs of
 * OvercookedGridworld.__init__, mdp.py:1090-1148, and Recipe.configure, mdp.py:221-336).
 * Filled from Python (oracle/oracle.py) via ctypes.  A recipe is identified by its ingredient
 * multiset (Recipe.ingredients is sorted, mdp.py:126-128) -> (n_onion, n_tomato).
 * ------------------------------------------------------------------------------------------ */
typedef struct OracleMdp {
    int32_t width, height;
    int32_t n_players;
    int32_t start_x[2], start_y[2];
    int32_t old_dynamics;
    int32_t max_num_ingredients; /* Recipe.MAX_NUM_INGREDIENTS, mdp.py:19,225 */
    int32_t n_all_orders;        /* 0 => every recipe is an order (OvercookedState.all_orders, mdp.py:881-887) */
    int32_t all_orders[MAX_RECIPES][2];
    int32_t n_bonus_orders;
    int32_t bonus_orders[MAX_RECIPES][2];
    int32_t has_cook_time, has_delivery_reward, has_recipe_values, has_recipe_times;
    int32_t has_ingredient_value, has_ingredient_time;
    double cook_time, delivery_reward;
    double recipe_values[MAX_RECIPES]; /* aligned with all_orders (Recipe.configure zip, mdp.py:310-316) */
    double recipe_times[MAX_RECIPES];
    double onion_value, tomato_value, onion_time, tomato_time;
    double order_bonus;
    double rew_placement_in_pot, rew_dish_pickup, rew_soup_pickup; /* mdp.py:1018-1025 */
    char terrain[MAX_CELLS]; /* terrain_mtx[y][x] at y*width+x, player digits already replaced by ' ' (mdp.py:1193-1206) */
} OracleMdp;

typedef struct Obj {
    int name;
    int n_ing;
    int ing[3]; /* NAME_ONION / NAME_TOMATO in insertion order (SoupState._ingredients, mdp.py:453) */
    int tick;   /* SoupState._cooking_tick, -1 idle (mdp.py:454,543-544) */
} Obj;

typedef struct Player {
    int present;
    int x, y;
    int o; /* Direction index */
    Obj held;
} Player;

typedef struct State {
    Player players[2];
    Obj objects[MAX_CELLS]; /* OvercookedState.objects: dict pos -> obj (mdp.py:798-811); name==NONE => key absent */
    int timestep;
} State;

/* ---------------------------------- Recipe ----------------------------------------------- */

static void count_ing(const Obj* soup, int* n_o, int* n_t) {
    int o = 0, t = 0;
    for (int i = 0; i < soup->n_ing; ++i) {
        if (soup->ing[i] == NAME_ONION) ++o;
        else ++t;
    }
    *n_o = o;
    *n_t = t;
}

static int find_order(const int32_t (*orders)[2], int n, int n_o, int n_t) {
    for (int i = 0; i < n; ++i)
        if (orders[i][0] == n_o && orders[i][1] == n_t) return i;
    return -1;
}

/* Recipe.value, mdp.py:136-161.  Python truthiness is restated literally: a configured value of 0
 * falls through to the next rule (`if self._delivery_reward:` / `if self._onion_value and self._tomato_value`). */
static double recipe_value(const OracleMdp* m, int n_o, int n_t) {
    if (m->has_delivery_reward && m->delivery_reward != 0.0) return m->delivery_reward;
    if (m->has_recipe_values && m->n_all_orders > 0) {
        int i = find_order(m->all_orders, m->n_all_orders, n_o, n_t);
        if (i >= 0) return m->recipe_values[i];
    }
    if (m->has_ingredient_value && m->onion_value != 0.0 && m->tomato_value != 0.0)
        return m->tomato_value * n_t + m->onion_value * n_o;
    return 20.0;
}

/* Recipe.time, mdp.py:163-188 */
static double recipe_time(const OracleMdp* m, int n_o, int n_t) {
    if (m->has_cook_time && m->cook_time != 0.0) return m->cook_time;
    if (m->has_recipe_times && m->n_all_orders > 0) {
        int i = find_order(m->all_orders, m->n_all_orders, n_o, n_t);
        if (i >= 0) return m->recipe_times[i];
    }
    if (m->has_ingredient_time && m->onion_time != 0.0 && m->tomato_time != 0.0)
        return m->onion_time * n_o + m->tomato_time * n_t;
    return 20.0;
}

/* get_recipe_value, non-discounted branch, mdp.py:1595-1602 */
static double get_recipe_value(const OracleMdp* m, int n_o, int n_t) {
    int in_all = (m->n_all_orders == 0) ? 1 : (find_order(m->all_orders, m->n_all_orders, n_o, n_t) >= 0);
    if (!in_all) return 0.0;
    if (find_order(m->bonus_orders, m->n_bonus_orders, n_o, n_t) < 0) return recipe_value(m, n_o, n_t);
    return m->order_bonus * recipe_value(m, n_o, n_t);
}

/* ---------------------------------- SoupState -------------------------------------------- */

static int soup_is_idle(const Obj* s) { return s->tick < 0; } /* mdp.py:543-544 */
static double soup_cook_time(const OracleMdp* m, const Obj* s) { /* mdp.py:525-530 -> recipe.time */
    int n_o, n_t;
    count_ing(s, &n_o, &n_t);
    return recipe_time(m, n_o, n_t);
}
static int soup_is_ready(const OracleMdp* m, const Obj* s) { /* mdp.py:537-540 */
    if (soup_is_idle(s)) return 0;
    return (double)s->tick >= soup_cook_time(m, s);
}
static int soup_is_cooking(const OracleMdp* m, const Obj* s) { /* mdp.py:507-508 */
    return !soup_is_idle(s) && !soup_is_ready(m, s);
}
static int soup_is_full(const OracleMdp* m, const Obj* s) { /* mdp.py:547-551 */
    return !soup_is_idle(s) || s->n_ing == m->max_num_ingredients;
}

/* ---------------------------------- helpers ---------------------------------------------- */

static char terrain_at(const OracleMdp* m, int x, int y) { return m->terrain[y * m->width + x]; } /* mdp.py:1783-1785 */
static int cell_of(const OracleMdp* m, int x, int y) { return y * m->width + x; }

enum { POT_EMPTY = 0, POT_1 = 1, POT_2 = 2, POT_3 = 3, POT_COOKING = 4, POT_READY = 5 };

typedef struct PotStates {
    int n_pots;
    int cls[MAX_CELLS]; /* class per pot, in get_pot_locations order */
} PotStates;

/* get_pot_states, mdp.py:1809-1838 */
static void get_pot_states(const OracleMdp* m, const State* s, PotStates* ps) {
    ps->n_pots = 0;
    for (int c = 0; c < m->width * m->height; ++c) {
        if (m->terrain[c] != 'P') continue;
        const Obj* soup = &s->objects[c];
        int cls;
        if (soup->name == NAME_NONE) cls = POT_EMPTY;
        else if (soup_is_ready(m, soup)) cls = POT_READY;
        else if (soup_is_cooking(m, soup)) cls = POT_COOKING;
        else cls = soup->n_ing; /* "{}_items" */
        ps->cls[ps->n_pots++] = cls;
    }
}

/* is_dish_pickup_useful, mdp.py:2180-2204 */
static int is_dish_pickup_useful(const OracleMdp* m, const State* s, const PotStates* ps) {
    if (m->n_players != 2) return 0;
    int dishes_on_counters = 0; /* get_counter_objects_dict(state)["dish"], mdp.py:1840-1851 */
    for (int c = 0; c < m->width * m->height; ++c)
        if (m->terrain[c] == 'X' && s->objects[c].name == NAME_DISH) ++dishes_on_counters;
    int num_player_dishes = 0; /* player_objects_by_type["dish"], mdp.py:851-862 */
    for (int p = 0; p < 2; ++p)
        if (s->players[p].present && s->players[p].held.name == NAME_DISH) ++num_player_dishes;
    int non_empty_pots = 0; /* ready + cooking + partially full (range(1, MAX), mdp.py:1882-1890) */
    for (int i = 0; i < ps->n_pots; ++i) {
        int c = ps->cls[i];
        if (c == POT_READY || c == POT_COOKING || (c >= 1 && c < m->max_num_ingredients)) ++non_empty_pots;
    }
    return dishes_on_counters == 0 && num_player_dishes < non_empty_pots;
}

/* ---------------------------------- event logging, mdp.py:1027-1058, 2121-2308 ------------- */

/* EVENT_TYPES order, mdp.py:1027-1058; bit index of (event k, player p) in the mask is 2*k + p */
enum {
    EV_TOMATO_PICKUP = 0, EV_USEFUL_TOMATO_PICKUP, EV_TOMATO_DROP, EV_USEFUL_TOMATO_DROP, EV_POTTING_TOMATO,
    EV_ONION_PICKUP, EV_USEFUL_ONION_PICKUP, EV_ONION_DROP, EV_USEFUL_ONION_DROP, EV_POTTING_ONION,
    EV_DISH_PICKUP, EV_USEFUL_DISH_PICKUP, EV_DISH_DROP, EV_USEFUL_DISH_DROP,
    EV_SOUP_PICKUP, EV_SOUP_DELIVERY, EV_SOUP_DROP,
    EV_OPTIMAL_ONION_POTTING, EV_OPTIMAL_TOMATO_POTTING, EV_VIABLE_ONION_POTTING, EV_VIABLE_TOMATO_POTTING,
    EV_CATASTROPHIC_ONION_POTTING, EV_CATASTROPHIC_TOMATO_POTTING, EV_USELESS_ONION_POTTING, EV_USELESS_TOMATO_POTTING
};

static void set_event(uint64_t* ev, int k, int player) { *ev |= (uint64_t)1 << (2 * k + player); }

/* get_full_pots: cooking + ready + "3_items", mdp.py:1872-1880 */
static int num_full_pots(const OracleMdp* m, const PotStates* ps) {
    int n = 0;
    for (int i = 0; i < ps->n_pots; ++i)
        if (ps->cls[i] == POT_COOKING || ps->cls[i] == POT_READY || ps->cls[i] == m->max_num_ingredients) ++n;
    return n;
}

static int other_holds(const State* s, int player_index, int name) {
    const Player* o = &s->players[1 - player_index];
    return o->present && o->held.name == name;
}

/* mdp.py:2223-2237 */
static int is_ingredient_pickup_useful(const OracleMdp* m, const State* s, const PotStates* ps, int player_index) {
    if (m->n_players != 2) return 0;
    int all_pots_full = ps->n_pots == num_full_pots(m, ps);
    return !(all_pots_full && !other_holds(s, player_index, NAME_DISH));
}
/* mdp.py:2239-2254 */
static int is_ingredient_drop_useful(const OracleMdp* m, const State* s, const PotStates* ps, int player_index) {
    if (m->n_players != 2) return 0;
    int all_pots_full = num_full_pots(m, ps) == ps->n_pots;
    return all_pots_full && !other_holds(s, player_index, NAME_DISH);
}
/* mdp.py:2206-2221 */
static int is_dish_drop_useful(const OracleMdp* m, const State* s, const PotStates* ps, int player_index) {
    if (m->n_players != 2) return 0;
    int all_non_full = num_full_pots(m, ps) == 0;
    return all_non_full && !other_holds(s, player_index, NAME_ONION);
}

/* log_object_pickup, mdp.py:2142-2159 */
static void log_object_pickup(const OracleMdp* m, uint64_t* ev, const State* s, int name, const PotStates* ps, int pi) {
    static const int key[5] = {-1, EV_ONION_PICKUP, EV_TOMATO_PICKUP, EV_DISH_PICKUP, EV_SOUP_PICKUP};
    static const int useful_key[5] = {-1, EV_USEFUL_ONION_PICKUP, EV_USEFUL_TOMATO_PICKUP, EV_USEFUL_DISH_PICKUP, -1};
    set_event(ev, key[name], pi);
    int useful = 0;
    if (name == NAME_ONION || name == NAME_TOMATO) useful = is_ingredient_pickup_useful(m, s, ps, pi);
    else if (name == NAME_DISH) useful = is_dish_pickup_useful(m, s, ps);
    if (useful) set_event(ev, useful_key[name], pi);
}

/* log_object_drop, mdp.py:2161-2178 */
static void log_object_drop(const OracleMdp* m, uint64_t* ev, const State* s, int name, const PotStates* ps, int pi) {
    static const int key[5] = {-1, EV_ONION_DROP, EV_TOMATO_DROP, EV_DISH_DROP, EV_SOUP_DROP};
    static const int useful_key[5] = {-1, EV_USEFUL_ONION_DROP, EV_USEFUL_TOMATO_DROP, EV_USEFUL_DISH_DROP, -1};
    set_event(ev, key[name], pi);
    int useful = 0;
    if (name == NAME_ONION || name == NAME_TOMATO) useful = is_ingredient_drop_useful(m, s, ps, pi);
    else if (name == NAME_DISH) useful = is_dish_drop_useful(m, s, ps, pi);
    if (useful) set_event(ev, useful_key[name], pi);
}

/* Value of the best recipe reachable from (n_o, n_t) by adding ingredients: the DFS of
 * _get_optimal_possible_recipe (mdp.py:1976-2016) followed by get_recipe_value of its result.  (0,0) stands
 * for the empty recipe (recipe == None). */
static double optimal_possible_value(const OracleMdp* m, int n_o, int n_t) {
    double best = 0.0;
    if (n_o + n_t >= 1) {
        double v = get_recipe_value(m, n_o, n_t);
        if (v > best) best = v;
    }
    if (n_o + n_t < m->max_num_ingredients) { /* Recipe.neighbors, mdp.py:193-205 */
        double a = optimal_possible_value(m, n_o + 1, n_t), b = optimal_possible_value(m, n_o, n_t + 1);
        if (a > best) best = a;
        if (b > best) best = b;
    }
    return best;
}

/* log_object_potting, mdp.py:2121-2140 with is_potting_{optimal,viable,catastrophic,useless} 2256-2308 */
static void log_object_potting(const OracleMdp* m, uint64_t* ev, const Obj* old_soup, const Obj* new_soup, int name, int pi) {
    int tomato = name == NAME_TOMATO;
    set_event(ev, tomato ? EV_POTTING_TOMATO : EV_POTTING_ONION, pi);
    int o_o, o_t, n_o, n_t;
    count_ing(old_soup, &o_o, &o_t);
    count_ing(new_soup, &n_o, &n_t);
    double old_val = optimal_possible_value(m, o_o, o_t);
    double new_val = optimal_possible_value(m, n_o, n_t);
    if (old_val == new_val) set_event(ev, EV_OPTIMAL_ONION_POTTING + tomato, pi);
    if (old_val > 0 && new_val == 0) set_event(ev, EV_CATASTROPHIC_ONION_POTTING + tomato, pi);
    if (new_val > 0) set_event(ev, EV_VIABLE_ONION_POTTING + tomato, pi);
    if (old_val == 0) set_event(ev, EV_USELESS_ONION_POTTING + tomato, pi);
}

/* ---------------------------------- resolve_interacts, mdp.py:1432-1579 ------------------- */

static void resolve_interacts(const OracleMdp* m, State* ns, const int* joint_action, double* sparse, double* shaped,
                              uint64_t* ev) {
    PotStates pot_states;
    get_pot_states(m, ns, &pot_states); /* once, before any interact: mdp.py:1439 */
    sparse[0] = sparse[1] = 0.0;
    shaped[0] = shaped[1] = 0.0;

    for (int player_idx = 0; player_idx < m->n_players; ++player_idx) {
        Player* player = &ns->players[player_idx];
        if (joint_action[player_idx] != A_INTERACT) continue;

        int ix = player->x + DIR_DX[player->o], iy = player->y + DIR_DY[player->o]; /* mdp.py:1452-1453 */
        char terrain_type = terrain_at(m, ix, iy);
        int i_pos = cell_of(m, ix, iy);
        Obj* at = &ns->objects[i_pos];
        int has_obj = player->held.name != NAME_NONE;

        if (terrain_type == 'X') {
            if (has_obj && at->name == NAME_NONE) { /* drop on counter, mdp.py:1459-1471 */
                log_object_drop(m, ev, ns, player->held.name, &pot_states, player_idx);
                *at = player->held;
                player->held.name = NAME_NONE;
            } else if (!has_obj && at->name != NAME_NONE) { /* pick up from counter, mdp.py:1473-1485 */
                log_object_pickup(m, ev, ns, at->name, &pot_states, player_idx);
                player->held = *at;
                at->name = NAME_NONE;
            }
        } else if (terrain_type == 'O' && !has_obj) { /* mdp.py:1487-1494 */
            log_object_pickup(m, ev, ns, NAME_ONION, &pot_states, player_idx);
            memset(&player->held, 0, sizeof(Obj));
            player->held.name = NAME_ONION;
        } else if (terrain_type == 'T' && !has_obj) { /* mdp.py:1496-1498 */
            memset(&player->held, 0, sizeof(Obj));
            player->held.name = NAME_TOMATO;
        } else if (terrain_type == 'D' && !has_obj) { /* mdp.py:1500-1513 */
            log_object_pickup(m, ev, ns, NAME_DISH, &pot_states, player_idx);
            if (is_dish_pickup_useful(m, ns, &pot_states)) shaped[player_idx] += m->rew_dish_pickup;
            memset(&player->held, 0, sizeof(Obj));
            player->held.name = NAME_DISH;
        } else if (terrain_type == 'P' && !has_obj) { /* mdp.py:1515-1522 */
            /* soup_to_be_cooked_at_location, mdp.py:1899-1908 */
            if (!m->old_dynamics && at->name == NAME_SOUP && !soup_is_cooking(m, at) && !soup_is_ready(m, at) &&
                at->n_ing > 0)
                at->tick = 0; /* begin_cooking, mdp.py:592-599 */
        } else if (terrain_type == 'P' && has_obj) {
            if (player->held.name == NAME_DISH && at->name == NAME_SOUP && soup_is_ready(m, at)) {
                /* soup pickup, mdp.py:1525-1539 */
                log_object_pickup(m, ev, ns, NAME_SOUP, &pot_states, player_idx);
                player->held = *at;
                at->name = NAME_NONE;
                shaped[player_idx] += m->rew_soup_pickup;
            } else if (player->held.name == NAME_ONION || player->held.name == NAME_TOMATO) {
                /* mdp.py:1541-1568 */
                if (at->name == NAME_NONE) {
                    memset(at, 0, sizeof(Obj));
                    at->name = NAME_SOUP;
                    at->tick = -1;
                }
                if (!soup_is_full(m, at)) {
                    Obj old_soup = *at; /* soup.deepcopy(), mdp.py:1551 */
                    int ing_name = player->held.name;
                    at->ing[at->n_ing++] = ing_name; /* add_ingredient, mdp.py:571-577 */
                    player->held.name = NAME_NONE;
                    shaped[player_idx] += m->rew_placement_in_pot;
                    log_object_potting(m, ev, &old_soup, at, ing_name, player_idx);
                }
            }
        } else if (terrain_type == 'S' && has_obj) { /* mdp.py:1570-1577 */
            if (player->held.name == NAME_SOUP) {
                int n_o, n_t;
                count_ing(&player->held, &n_o, &n_t);
                player->held.name = NAME_NONE;                /* deliver_soup, mdp.py:1631-1642 */
                sparse[player_idx] += get_recipe_value(m, n_o, n_t);
                set_event(ev, EV_SOUP_DELIVERY, player_idx);
            }
        }
    }
}

/* ---------------------------------- resolve_movement, mdp.py:1644-1727 -------------------- */

static void resolve_movement(const OracleMdp* m, State* s, const int* joint_action) {
    int old_x[2], old_y[2], new_x[2], new_y[2], new_o[2];
    int np = m->n_players;
    for (int i = 0; i < np; ++i) {
        const Player* p = &s->players[i];
        int a = joint_action[i];
        old_x[i] = p->x;
        old_y[i] = p->y;
        /* _move_if_direction, mdp.py:1718-1727 */
        if (a == A_INTERACT) {
            new_x[i] = p->x; new_y[i] = p->y; new_o[i] = p->o;
        } else {
            int nx = p->x + DIR_DX[a], ny = p->y + DIR_DY[a];
            new_o[i] = (a == A_STAY) ? p->o : a;
            if (terrain_at(m, nx, ny) != ' ') { nx = p->x; ny = p->y; }
            new_x[i] = nx; new_y[i] = ny;
        }
    }
    /* is_transition_collision, mdp.py:1673-1683 */
    int collision = 0;
    if (np == 2) {
        if (new_x[0] == new_x[1] && new_y[0] == new_y[1]) collision = 1;
        if (new_x[0] == old_x[1] && new_y[0] == old_y[1] && old_x[0] == new_x[1] && old_y[0] == new_y[1]) collision = 1;
    }
    for (int i = 0; i < np; ++i) {
        Player* p = &s->players[i];
        if (!collision) { p->x = new_x[i]; p->y = new_y[i]; } /* _handle_collisions, mdp.py:1705-1709 */
        p->o = new_o[i];                                       /* orientation always updated, mdp.py:1652-1655 */
    }
}

/* ---------------------------------- step_environment_effects, mdp.py:1691-1703 ------------ */

static void step_environment_effects(const OracleMdp* m, State* s) {
    s->timestep += 1;
    for (int c = 0; c < m->width * m->height; ++c) {
        Obj* obj = &s->objects[c];
        if (obj->name != NAME_SOUP) continue;
        if (m->old_dynamics && !soup_is_cooking(m, obj) && !soup_is_ready(m, obj) && obj->n_ing == 3) obj->tick = 0;
        if (soup_is_cooking(m, obj)) obj->tick += 1;
    }
}

/* get_state_transition, mdp.py:1375-1430 */
static void state_transition(const OracleMdp* m, State* ns, const int* joint_action, double* sparse, double* shaped,
                             uint64_t* ev) {
    resolve_interacts(m, ns, joint_action, sparse, shaped, ev);
    resolve_movement(m, ns, joint_action);
    step_environment_effects(m, ns);
}

/* get_standard_start_state, mdp.py:1297-1305 + from_player_positions 939-950 */
static void standard_start_state(const OracleMdp* m, State* s) {
    memset(s, 0, sizeof(State));
    for (int i = 0; i < m->n_players; ++i) {
        s->players[i].present = 1;
        s->players[i].x = m->start_x[i];
        s->players[i].y = m->start_y[i];
        s->players[i].o = A_NORTH;
    }
}

/* ---------------------------------- wire format (include/oc_amd.h) ------------------------ */

static int obj_to_code(const Obj* o) {
    switch (o->name) {
        case NAME_NONE: return 0;
        case NAME_ONION: return 1;
        case NAME_TOMATO: return 2;
        case NAME_DISH: return 3;
        default: {
            int bits = 0;
            for (int i = 0; i < o->n_ing; ++i)
                if (o->ing[i] == NAME_TOMATO) bits |= 1 << i;
            return 0x80 | (o->n_ing << 3) | bits;
        }
    }
}

static void code_to_obj(int code, Obj* o) {
    memset(o, 0, sizeof(Obj));
    if (code == 0) return;
    if (code < 0x80) { o->name = code; return; }
    o->name = NAME_SOUP;
    o->n_ing = (code >> 3) & 3;
    for (int i = 0; i < o->n_ing; ++i) o->ing[i] = ((code >> i) & 1) ? NAME_TOMATO : NAME_ONION;
    o->tick = -1;
}

static int n_obj_planes(const OracleMdp* m) { return (m->width * m->height + 15) / 16; }

static void unpack_state(const OracleMdp* m, const uint8_t* planes, int64_t n_envs, int64_t e, State* s) {
    const uint8_t* hdr = planes + 16 * e;
    memset(s, 0, sizeof(State));
    int ncells = m->width * m->height;
    for (int c = 0; c < ncells; ++c) {
        const uint8_t* pl = planes + 16 * ((int64_t)(1 + (c >> 4)) * n_envs + e);
        code_to_obj(pl[c & 15], &s->objects[c]);
    }
    /* pot ticks: slot k = k-th 'P' cell in row-major order (get_pot_locations order, mdp.py:1711-1716,1799) */
    int slot = 0;
    for (int c = 0; c < ncells; ++c) {
        if (m->terrain[c] != 'P') continue;
        if (s->objects[c].name == NAME_SOUP && slot < MAX_POTS) s->objects[c].tick = (int)hdr[8 + slot] - 1;
        ++slot;
    }
    for (int c = 0; c < ncells; ++c) { /* soups outside pots are cooked: tick = cook time */
        Obj* o = &s->objects[c];
        if (o->name == NAME_SOUP && m->terrain[c] != 'P') o->tick = (int)soup_cook_time(m, o);
    }
    for (int p = 0; p < 2; ++p) {
        Player* pl = &s->players[p];
        int pos = hdr[3 * p];
        if (pos == 0xFF) { pl->present = 0; continue; }
        pl->present = 1;
        pl->x = pos % m->width;
        pl->y = pos / m->width;
        pl->o = hdr[3 * p + 1];
        code_to_obj(hdr[3 * p + 2], &pl->held);
        if (pl->held.name == NAME_SOUP) pl->held.tick = (int)soup_cook_time(m, &pl->held);
    }
    s->timestep = hdr[6] | (hdr[7] << 8);
}

static void pack_state(const OracleMdp* m, const State* s, uint8_t* planes, int64_t n_envs, int64_t e) {
    uint8_t* hdr = planes + 16 * e;
    memset(hdr, 0, 16);
    int ncells = m->width * m->height;
    for (int pl = 0; pl < n_obj_planes(m); ++pl) memset(planes + 16 * ((int64_t)(1 + pl) * n_envs + e), 0, 16);
    for (int c = 0; c < ncells; ++c) {
        uint8_t* pl = planes + 16 * ((int64_t)(1 + (c >> 4)) * n_envs + e);
        pl[c & 15] = (uint8_t)obj_to_code(&s->objects[c]);
    }
    int slot = 0;
    for (int c = 0; c < ncells; ++c) {
        if (m->terrain[c] != 'P') continue;
        if (s->objects[c].name == NAME_SOUP && slot < MAX_POTS) hdr[8 + slot] = (uint8_t)(s->objects[c].tick + 1);
        ++slot;
    }
    for (int p = 0; p < 2; ++p) {
        const Player* pl = &s->players[p];
        if (!pl->present) { hdr[3 * p] = 0xFF; continue; }
        hdr[3 * p] = (uint8_t)(pl->y * m->width + pl->x);
        hdr[3 * p + 1] = (uint8_t)pl->o;
        hdr[3 * p + 2] = (uint8_t)obj_to_code(&pl->held);
    }
    hdr[6] = (uint8_t)(s->timestep & 0xFF);
    hdr[7] = (uint8_t)((s->timestep >> 8) & 0xFF);
}

/* ---------------------------------- batch entry points ------------------------------------ */

#define F_DONE 0x01
#define F_BAD_ACTION 0x02
#define F_RESET 0x04
#define OPT_AUTO_RESET 0x1u

/* start_state_fn of a batch: NULL = get_standard_start_state, else get_random_start_state_fn(random_start_pos,
 * rnd_obj_prob_thresh) (mdp.py:1307-1369) with the draws of oc_reset_random (include/oc_amd.h, OcStartSpec) */
typedef struct OracleStartSpec {
    uint64_t seed;
    int64_t env_offset;
    uint32_t epoch;
    int32_t random_start_pos;
    double rnd_obj_prob_thresh;
    /* regen_count > 0: OvercookedEnv.reset(regen_mdp=True) over a generator that yields another layout every episode
     * (env.py:288-302): a restarting env first moves to layout regen_first + draw % regen_count (include/oc_amd.h) */
    uint32_t regen_first, regen_count;
} OracleStartSpec;

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
static uint32_t draw_layout_id(const OracleStartSpec* ss, uint64_t g, uint32_t epoch) {
    uint32_t ctr[4] = {epoch, (uint32_t)g, (uint32_t)(g >> 32), 15u};
    uint32_t key[2] = {(uint32_t)ss->seed, (uint32_t)(ss->seed >> 32) ^ 0x52535421u}, r[4];
    oracle_philox4x32_10(ctr, key, r);
    return ss->regen_first + (uint32_t)(((uint64_t)r[0] * ss->regen_count) >> 32);
}

static void random_start_state(const OracleMdp* m, State* s, uint64_t seed, uint64_t g, uint32_t epoch,
                               int random_start_pos, uint64_t thresh);

/* mdps / lid: the layout table and this env's (writable) layout id, for restarts that re-draw the layout; *mp is the env's
 * current mdp and follows such a move */
static void env_step_one(const OracleMdp** mp, State* s, const int* ja, float* rew4, uint8_t* flag, float* ep4,
                         int horizon, uint32_t options, uint64_t* events, const OracleStartSpec* ss, uint64_t g,
                         uint32_t epoch, const OracleMdp* mdps, uint16_t* lid) {
    const OracleMdp* m = *mp;
    double sparse[2], shaped[2];
    uint8_t f = 0;
    uint64_t ev = 0;
    if (events) *events = 0;
    if (ja[0] < 0 || ja[0] > 5 || ja[1] < 0 || ja[1] > 5) { /* mdp.py:1394-1398 raises ValueError */
        if (rew4) rew4[0] = rew4[1] = rew4[2] = rew4[3] = 0.f;
        if (flag) *flag = F_BAD_ACTION;
        return;
    }
    state_transition(m, s, ja, sparse, shaped, &ev);
    if (events) *events = ev;
    if (rew4) {
        rew4[0] = (float)sparse[0]; rew4[1] = (float)sparse[1];
        rew4[2] = (float)shaped[0]; rew4[3] = (float)shaped[1];
    }
    if (ep4) { /* _update_game_stats, env.py:387-392 */
        ep4[0] += (float)sparse[0]; ep4[1] += (float)sparse[1];
        ep4[2] += (float)shaped[0]; ep4[3] += (float)shaped[1];
    }
    if (s->timestep >= horizon) { /* is_done, env.py:321-325 */
        f |= F_DONE;
        if (options & OPT_AUTO_RESET) { /* OvercookedEnv.reset, env.py:288-319: start_state_fn() or the standard state */
            if (ss && ss->regen_count && lid) { /* regen_mdp: the next episode's layout (env.py:293-302) */
                *lid = (uint16_t)draw_layout_id(ss, g, epoch);
                m = *mp = &mdps[*lid];
            }
            if (ss)
                random_start_state(m, s, ss->seed, g, epoch, ss->random_start_pos,
                                   (uint64_t)(ss->rnd_obj_prob_thresh * 4294967296.0));
            else
                standard_start_state(m, s);
            if (ep4) ep4[0] = ep4[1] = ep4[2] = ep4[3] = 0.f;
            f |= F_RESET;
        }
    }
    if (flag) *flag = f;
}

int oracle_step(const OracleMdp* mdps, int n_mdps, uint16_t* layout_id, const uint8_t* state_in,
                uint8_t* state_out, const uint8_t* actions, float* rewards, uint8_t* flags, float* ep_returns,
                uint64_t* events, int64_t n_envs, int horizon, uint32_t options, const OracleStartSpec* ss) {
    (void)n_mdps;
    for (int64_t e = 0; e < n_envs; ++e) {
        const OracleMdp* m = &mdps[layout_id ? layout_id[e] : 0];
        State s;
        unpack_state(m, state_in, n_envs, e, &s);
        int ja[2] = {actions[2 * e], actions[2 * e + 1]};
        env_step_one(&m, &s, ja, rewards ? rewards + 4 * e : 0, flags ? flags + e : 0, ep_returns ? ep_returns + 4 * e : 0,
                     horizon, options, events ? events + e : 0, ss, ss ? (uint64_t)(ss->env_offset + e) : 0,
                     ss ? ss->epoch : 0, mdps, layout_id ? layout_id + e : 0);
        pack_state(m, &s, state_out, n_envs, e);
    }
    return 0;
}

/* oc_regen_layouts: new layout ids for the envs an explicit reset is about to restart */
int oracle_regen_layouts(uint16_t* layout_id, const uint8_t* mask, uint8_t mask_bits, int64_t n_envs, const OracleStartSpec* ss) {
    if (!layout_id || !ss || !ss->regen_count) return -1;
    for (int64_t e = 0; e < n_envs; ++e) {
        if (mask && !(mask[e] & mask_bits)) continue;
        layout_id[e] = (uint16_t)draw_layout_id(ss, (uint64_t)(ss->env_offset + e), ss->epoch);
    }
    return 0;
}

int oracle_reset(const OracleMdp* mdps, int n_mdps, const uint16_t* layout_id, uint8_t* state, const uint8_t* mask,
                 float* ep_returns, int64_t n_envs) {
    (void)n_mdps;
    for (int64_t e = 0; e < n_envs; ++e) {
        if (mask && !mask[e]) continue;
        const OracleMdp* m = &mdps[layout_id ? layout_id[e] : 0];
        State s;
        standard_start_state(m, &s);
        pack_state(m, &s, state, n_envs, e);
        if (ep_returns) memset(ep_returns + 4 * e, 0, 4 * sizeof(float));
    }
    return 0;
}

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123).
 * Not part of the reference: it stands in for the reference's RandomAgent(all_actions=True)
 * (agents/agent.py:223-260, np.random.choice over the 6 actions) so that rollouts are reproducible
 * on both sides.  Checked against the Random123 known-answer vectors in tests/test_oracle_golden.py. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Action stream of oc_rollout_random (include/oc_amd.h): one Philox block feeds 8 consecutive steps; word s/2
 * of block t/8 is expanded into base-6 digits by multiply-high, two digits (player 0, player 1) per step. */
static void draw_actions(uint64_t seed, uint64_t g, uint64_t t, int* a0, int* a1) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t blk = t >> 3;
    uint32_t s8 = (uint32_t)(t & 7u);
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(blk >> 32)};
    uint32_t r[4];
    oracle_philox4x32_10(ctr, key, r);
    uint32_t x = r[s8 >> 1];
    if (s8 & 1u) x *= 36u;
    *a0 = (int)(((uint64_t)x * 6u) >> 32);
    x *= 6u;
    *a1 = (int)(((uint64_t)x * 6u) >> 32);
}

void oracle_random_actions(uint64_t seed, int64_t env_offset, int64_t t, int64_t n_envs, uint8_t* actions) {
    for (int64_t e = 0; e < n_envs; ++e) {
        int a0, a1;
        draw_actions(seed, (uint64_t)(env_offset + e), (uint64_t)t, &a0, &a1);
        actions[2 * e] = (uint8_t)a0;
        actions[2 * e + 1] = (uint8_t)a1;
    }
}

static int g_threads = 1;
int oracle_set_threads(int n) { /* returns the number of threads that will be used (1 without OpenMP) */
#ifdef _OPENMP
    g_threads = n > 0 ? n : 1;
#else
    (void)n;
    g_threads = 1;
#endif
    return g_threads;
}

int oracle_rollout_random(const OracleMdp* mdps, int n_mdps, uint16_t* layout_id, uint8_t* state, float* rewards,
                          uint8_t* flags, float* ep_returns, int64_t n_envs, int horizon, uint32_t options,
                          uint64_t seed, int64_t env_offset, int64_t t0, int n_steps, const OracleStartSpec* ss) {
    (void)n_mdps;
    /* envs are independent (no cross-env data flow in mdp.py): with OpenMP the env loop spreads over the host cores set
     * by oracle_set_threads (default 1) — used only by bench.py's cpu_baseline to report an all-cores figure */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads)
#endif
    for (int64_t e = 0; e < n_envs; ++e) {
        const OracleMdp* m = &mdps[layout_id ? layout_id[e] : 0];
        State s;
        unpack_state(m, state, n_envs, e, &s);
        uint64_t g = (uint64_t)(env_offset + e);
        for (int k = 0; k < n_steps; ++k) {
            int ja[2];
            draw_actions(seed, g, (uint64_t)(t0 + k), &ja[0], &ja[1]);
            env_step_one(&m, &s, ja, rewards ? rewards + 4 * ((int64_t)k * n_envs + e) : 0,
                         flags ? flags + ((int64_t)k * n_envs + e) : 0, ep_returns ? ep_returns + 4 * e : 0, horizon,
                         options, 0, ss, g, ss ? ss->epoch + (uint32_t)k : 0, /* restart at step k: epoch + k */
                         mdps, layout_id ? layout_id + e : 0);
        }
        pack_state(m, &s, state, n_envs, e);
    }
    return 0;
}

/* ---------------------------------- lossless_state_encoding, mdp.py:2385-2561 ------------- */

/* obs: [n_envs][2][W][H][26] int32 here (the reference emits int64 via astype(int), mdp.py:2554). */
static void encode_one(const OracleMdp* m, const State* st, int horizon, int32_t* obs) {
    int W = m->width, H = m->height;
    size_t per_player = (size_t)W * H * NUM_LAYERS;
    memset(obs, 0, 2 * per_player * sizeof(int32_t));
#define LAYER(base, x, y, l) (base)[((size_t)(x) * H + (y)) * NUM_LAYERS + (l)]
    for (int primary = 0; primary < 2; ++primary) {
        int32_t* o = obs + primary * per_player;
        int other = 1 - primary;
        /* urgency, mdp.py:2446-2447 */
        if (horizon - st->timestep < 40)
            for (int x = 0; x < W; ++x)
                for (int y = 0; y < H; ++y) LAYER(o, x, y, 25) = 1;
        /* base map layers 10..15, mdp.py:2449-2465 */
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                char t = terrain_at(m, x, y);
                if (t == 'P') LAYER(o, x, y, 10) = 1;
                if (t == 'X') LAYER(o, x, y, 11) = 1;
                if (t == 'O') LAYER(o, x, y, 12) = 1;
                if (t == 'T') LAYER(o, x, y, 13) = 1;
                if (t == 'D') LAYER(o, x, y, 14) = 1;
                if (t == 'S') LAYER(o, x, y, 15) = 1;
            }
        /* player layers, mdp.py:2468-2479 with the ordering of 2423-2434 */
        int order[2] = {primary, other};
        for (int k = 0; k < 2; ++k) {
            const Player* p = &st->players[order[k]];
            LAYER(o, p->x, p->y, k) = 1;
            LAYER(o, p->x, p->y, 2 + 4 * k + p->o) = 1;
        }
        /* object layers, mdp.py:2482-2534; all_objects_list includes held objects (876-879) */
        for (int idx = 0; idx < W * H + 2; ++idx) {
            const Obj* ob;
            int x, y;
            if (idx < W * H) {
                ob = &st->objects[idx];
                x = idx % W;
                y = idx / W;
            } else {
                const Player* p = &st->players[idx - W * H];
                ob = &p->held;
                x = p->x;
                y = p->y;
            }
            if (ob->name == NAME_NONE) continue;
            if (ob->name == NAME_SOUP) {
                int n_o, n_t;
                count_ing(ob, &n_o, &n_t);
                if (terrain_at(m, x, y) == 'P' && idx < W * H) {
                    if (soup_is_idle(ob)) {
                        LAYER(o, x, y, 16) += n_o;
                        LAYER(o, x, y, 17) += n_t;
                    } else {
                        LAYER(o, x, y, 18) += n_o;
                        LAYER(o, x, y, 19) += n_t;
                        LAYER(o, x, y, 20) += (int32_t)(soup_cook_time(m, ob) - ob->tick);
                        if (soup_is_ready(m, ob)) LAYER(o, x, y, 21) += 1;
                    }
                } else {
                    LAYER(o, x, y, 18) += n_o;
                    LAYER(o, x, y, 19) += n_t;
                    LAYER(o, x, y, 21) += 1;
                }
            } else if (ob->name == NAME_DISH) LAYER(o, x, y, 22) += 1;
            else if (ob->name == NAME_ONION) LAYER(o, x, y, 23) += 1;
            else if (ob->name == NAME_TOMATO) LAYER(o, x, y, 24) += 1;
        }
    }
#undef LAYER
}

int oracle_encode_lossless(const OracleMdp* mdps, int n_mdps, const uint16_t* layout_id, const uint8_t* state,
                           int32_t* obs, int64_t n_envs, int horizon) {
    (void)n_mdps;
    for (int64_t e = 0; e < n_envs; ++e) {
        const OracleMdp* m = &mdps[layout_id ? layout_id[e] : 0];
        State s;
        unpack_state(m, state, n_envs, e, &s);
        if (m->n_players != 2) return -1; /* assert, mdp.py:2389-2391 */
        encode_one(m, &s, horizon, obs + (size_t)e * 2 * m->width * m->height * NUM_LAYERS);
    }
    return 0;
}

/* The same encoding narrowed to bytes (every value of the 26 layers is in 0..255: ingredient counts, cook times <= 254,
 * 0/1 flags), env loop spread over oracle_set_threads cores: what bench.py's parity check of BASELINE configs[2] compares
 * with the u8 observations of EVERY step of a launch.  Returns -1 for a non-2-player layout, -2 for a value above 255. */
int oracle_encode_lossless_u8(const OracleMdp* mdps, int n_mdps, const uint16_t* layout_id, const uint8_t* state,
                              uint8_t* obs, int64_t n_envs, int horizon) {
    (void)n_mdps;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads)
#endif
    for (int64_t e = 0; e < n_envs; ++e) {
        const OracleMdp* m = &mdps[layout_id ? layout_id[e] : 0];
        int32_t tmp[2 * MAX_CELLS * NUM_LAYERS];
        State s;
        if (m->n_players != 2) { rc = -1; continue; }
        const size_t len = (size_t)2 * m->width * m->height * NUM_LAYERS;
        memset(tmp, 0, len * sizeof(int32_t));
        unpack_state(m, state, n_envs, e, &s);
        encode_one(m, &s, horizon, tmp);
        uint8_t* dst = obs + (size_t)e * len;
        for (size_t i = 0; i < len; ++i) {
            if (tmp[i] < 0 || tmp[i] > 255) rc = -2;
            dst[i] = (uint8_t)tmp[i];
        }
    }
    return rc;
}

size_t oracle_mdp_size(void) { return sizeof(OracleMdp); }

/* ====================================================================================================
 * featurize_state (mdp.py:2579-2898) with the MotionPlanner distances it needs
 * (planning/planners.py:46-450, planning/search.py:167-310).
 * ==================================================================================================== */

typedef struct Planner {
    int n_floor;
    int floor_cell[MAX_CELLS]; /* get_valid_player_positions(): free cells, row-major (mdp.py:1733-1734) */
    int floor_idx[MAX_CELLS];  /* cell -> index in floor_cell or -1 */
    int region[MAX_CELLS];     /* connected component of the free cell (Graph.are_in_same_cc, search.py:294-310) */
    int n_states;              /* (pos, orientation) states, planners.py:315-340 */
    int* dist;                 /* [n_states][n_states] shortest path lengths, -1 unreachable (search.py:205-213) */
    int counter_goal[MAX_CELLS]; /* MotionPlanner.counter_goals as a cell mask */
} Planner;

/* _move_if_direction (mdp.py:1718-1727) on (cell, orientation) states for a motion action 0..3 */
static int planner_successor(const OracleMdp* m, const Planner* pl, int state, int a) {
    int c = pl->floor_cell[state / 4];
    int x = c % m->width + DIR_DX[a], y = c / m->width + DIR_DY[a];
    int nc = y * m->width + x;
    if (m->terrain[nc] != ' ') return (state / 4) * 4 + a; /* blocked: turn in place */
    return pl->floor_idx[nc] * 4 + a;
}

static void planner_build(const OracleMdp* m, const int* counter_goal_mask, Planner* pl) {
    int ncells = m->width * m->height;
    pl->n_floor = 0;
    for (int c = 0; c < ncells; ++c) {
        pl->floor_idx[c] = -1;
        pl->region[c] = -1;
        pl->counter_goal[c] = counter_goal_mask ? counter_goal_mask[c] : 0;
        if (m->terrain[c] == ' ') { pl->floor_idx[c] = pl->n_floor; pl->floor_cell[pl->n_floor++] = c; }
    }
    /* connected components of the free region */
    int n_regions = 0, stack[MAX_CELLS];
    for (int i = 0; i < pl->n_floor; ++i) {
        int c0 = pl->floor_cell[i];
        if (pl->region[c0] >= 0) continue;
        int sp = 0;
        stack[sp++] = c0;
        pl->region[c0] = n_regions;
        while (sp) {
            int c = stack[--sp];
            for (int d = 0; d < 4; ++d) {
                int nc = (c / m->width + DIR_DY[d]) * m->width + c % m->width + DIR_DX[d];
                if (m->terrain[nc] == ' ' && pl->region[nc] < 0) { pl->region[nc] = n_regions; stack[sp++] = nc; }
            }
        }
        ++n_regions;
    }
    /* all-pairs BFS over (pos, orientation) states with unit action costs (planners.py:342-358) */
    pl->n_states = pl->n_floor * 4;
    int ns = pl->n_states;
    pl->dist = (int*)malloc(sizeof(int) * (size_t)ns * ns);
    int* queue = (int*)malloc(sizeof(int) * (size_t)ns);
    for (int s0 = 0; s0 < ns; ++s0) {
        int* d = pl->dist + (size_t)s0 * ns;
        for (int i = 0; i < ns; ++i) d[i] = -1;
        int qh = 0, qt = 0;
        d[s0] = 0;
        queue[qt++] = s0;
        while (qh < qt) {
            int s = queue[qh++];
            for (int a = 0; a < 4; ++a) {
                int t = planner_successor(m, pl, s, a);
                if (d[t] < 0) { d[t] = d[s] + 1; queue[qt++] = t; }
            }
        }
    }
    free(queue);
}

/* is_valid_motion_goal (planners.py:211-230) for the goal "stand on `adj`, face feature cell `f`" */
static int valid_goal_feature(const OracleMdp* m, const Planner* pl, int f) {
    char t = m->terrain[f];
    if (t == ' ') return 0;
    if (t == 'X' && !pl->counter_goal[f]) return 0;
    return 1;
}

/* min_cost_to_feature(..., with_argmin=True) (planners.py:391-423): returns the best feature cell or -1 */
static int min_cost_to_feature(const OracleMdp* m, const Planner* pl, int start_cell, int start_or, const int* features,
                               int n_features, int* min_cost_out) {
    int start = pl->floor_idx[start_cell] * 4 + start_or;
    int min_dist = -1, best = -1;
    for (int i = 0; i < n_features; ++i) {
        int f = features[i];
        int fx = f % m->width, fy = f / m->width;
        for (int d = 0; d < 4; ++d) { /* _get_possible_motion_goals_for_feature, planners.py:439-450 */
            int ax = fx + DIR_DX[d], ay = fy + DIR_DY[d];
            if (ax < 0 || ay < 0 || ax >= m->width || ay >= m->height) continue;
            int adj = ay * m->width + ax;
            if (m->terrain[adj] != ' ') continue;
            static const int OPP[4] = {1, 0, 3, 2}; /* OPPOSITE_DIRECTIONS, actions.py:17 */
            int goal = pl->floor_idx[adj] * 4 + OPP[d];
            if (!valid_goal_feature(m, pl, f)) continue;
            if (pl->region[adj] != pl->region[start_cell]) continue; /* positions_are_connected */
            int cur = pl->dist[(size_t)start * pl->n_states + goal];
            if (cur < 0) continue;
            if (min_dist < 0 || cur < min_dist) { best = f; min_dist = cur; }
        }
    }
    if (min_cost_out) *min_cost_out = min_dist + 1;
    return best;
}

/* one player's block of featurize_state: appends to `out`, returns the number of values written */
static int featurize_player(const OracleMdp* m, const Planner* pl, const State* st, int i, int num_pots, float* out) {
    const Player* p = &st->players[i];
    int ncells = m->width * m->height, k = 0;
    int pcell = p->y * m->width + p->x;
    /* orientation one-hot, held object one-hot over IDX_TO_OBJ = [onion, soup, dish, tomato] (mdp.py:2739-2765) */
    for (int d = 0; d < 4; ++d) out[k++] = (p->o == d) ? 1.f : 0.f;
    static const int OBJ_SLOT[5] = {-1, 0, 3, 2, 1};
    for (int j = 0; j < 4; ++j) out[k++] = (p->held.name != NAME_NONE && OBJ_SLOT[p->held.name] == j) ? 1.f : 0.f;
    /* closest onion / tomato / dish / soup / serving / empty counter (make_closest_feature, mdp.py:2622-2655) */
    static const char DISP[3] = {'O', 'T', 'D'};
    static const int NAMES[3] = {NAME_ONION, NAME_TOMATO, NAME_DISH};
    int feats[MAX_CELLS];
    for (int q = 0; q < 6; ++q) {
        int nf = 0, held_match = 0;
        if (q < 3) {
            for (int c = 0; c < ncells; ++c) if (m->terrain[c] == DISP[q]) feats[nf++] = c;           /* dispensers first */
            for (int c = 0; c < ncells; ++c)
                if (m->terrain[c] == 'X' && st->objects[c].name == NAMES[q]) feats[nf++] = c;          /* then counters */
            held_match = p->held.name == NAMES[q];
        } else if (q == 3) {
            for (int c = 0; c < ncells; ++c) if (m->terrain[c] == 'X' && st->objects[c].name == NAME_SOUP) feats[nf++] = c;
            held_match = p->held.name == NAME_SOUP;
        } else if (q == 4) {
            for (int c = 0; c < ncells; ++c) if (m->terrain[c] == 'S') feats[nf++] = c;
        } else {
            for (int c = 0; c < ncells; ++c) if (m->terrain[c] == 'X' && st->objects[c].name == NAME_NONE) feats[nf++] = c;
        }
        const Obj* obj = 0;
        float dx = 0.f, dy = 0.f;
        if (held_match) obj = &p->held;
        else {
            int loc = min_cost_to_feature(m, pl, pcell, p->o, feats, nf, 0);
            if (loc >= 0) {
                dx = (float)(loc % m->width - p->x);  /* pos_distance(location, player.position), utils.py:95 */
                dy = (float)(loc / m->width - p->y);
                if (st->objects[loc].name != NAME_NONE) obj = &st->objects[loc];
            }
        }
        out[k++] = dx; out[k++] = dy;
        if (q == 3) {
            int n_o = 0, n_t = 0;
            if (obj) count_ing(obj, &n_o, &n_t);
            out[k++] = (float)n_o; out[k++] = (float)n_t;
        }
    }
    /* the num_pots closest pots (mdp.py:2818-2829, make_pot_feature 2657-2731) */
    int pots[MAX_CELLS], np_ = 0;
    for (int c = 0; c < ncells; ++c) if (m->terrain[c] == 'P') pots[np_++] = c;
    for (int j = 0; j < num_pots; ++j) {
        int loc = min_cost_to_feature(m, pl, pcell, p->o, pots, np_, 0);
        if (loc < 0) { for (int z = 0; z < 10; ++z) out[k++] = 0.f; continue; }
        const Obj* soup = &st->objects[loc];
        int is_empty = soup->name == NAME_NONE;
        int is_ready = !is_empty && soup_is_ready(m, soup), is_cooking = !is_empty && soup_is_cooking(m, soup);
        int is_full = is_cooking || is_ready || (!is_empty && soup->n_ing == m->max_num_ingredients);
        int n_o = 0, n_t = 0;
        double remaining = 0;
        if (!is_empty) {
            count_ing(soup, &n_o, &n_t);
            if (!soup_is_idle(soup)) { remaining = soup_cook_time(m, soup) - soup->tick; if (remaining < 0) remaining = 0; }
        }
        out[k++] = 1.f; out[k++] = (float)is_empty; out[k++] = (float)is_full; out[k++] = (float)is_cooking;
        out[k++] = (float)is_ready; out[k++] = (float)n_o; out[k++] = (float)n_t; out[k++] = (float)remaining;
        out[k++] = (float)(loc % m->width - p->x); out[k++] = (float)(loc / m->width - p->y);
        int w = 0;
        for (int z = 0; z < np_; ++z) if (pots[z] != loc) pots[w++] = pots[z]; /* pot_locations.remove(closest) */
        np_ = w;
    }
    /* walls in the four directions (mdp.py:2831-2838) */
    for (int d = 0; d < 4; ++d) out[k++] = terrain_at(m, p->x + DIR_DX[d], p->y + DIR_DY[d]) == ' ' ? 0.f : 1.f;
    return k;
}

/* features: [n_envs][2][2*(num_pots*10 + 26) + 4] floats; counter_goal_mask: [n_mdps][MAX_CELLS] ints or NULL (none) */
int oracle_featurize(const OracleMdp* mdps, int n_mdps, const uint16_t* layout_id, const int32_t* counter_goal_mask,
                     const uint8_t* state, float* features, int64_t n_envs, int num_pots) {
    Planner* pls = (Planner*)calloc((size_t)n_mdps, sizeof(Planner));
    for (int l = 0; l < n_mdps; ++l) {
        int mask[MAX_CELLS];
        for (int c = 0; c < MAX_CELLS; ++c) mask[c] = counter_goal_mask ? counter_goal_mask[l * MAX_CELLS + c] : 0;
        planner_build(&mdps[l], mask, &pls[l]);
    }
    int per = num_pots * 10 + 26, total = 2 * per + 4;
    for (int64_t e = 0; e < n_envs; ++e) {
        int l = layout_id ? layout_id[e] : 0;
        const OracleMdp* m = &mdps[l];
        State s;
        unpack_state(m, state, n_envs, e, &s);
        float blocks[2][256];
        for (int i = 0; i < 2; ++i) featurize_player(m, &pls[l], &s, i, num_pots, blocks[i]);
        for (int i = 0; i < 2; ++i) { /* [own, other, other - own position, own position] (mdp.py:2849-2896) */
            float* o = features + ((size_t)e * 2 + i) * total;
            memcpy(o, blocks[i], per * sizeof(float));
            memcpy(o + per, blocks[1 - i], per * sizeof(float));
            o[2 * per + 0] = (float)(s.players[1 - i].x - s.players[i].x);
            o[2 * per + 1] = (float)(s.players[1 - i].y - s.players[i].y);
            o[2 * per + 2] = (float)s.players[i].x;
            o[2 * per + 3] = (float)s.players[i].y;
        }
    }
    for (int l = 0; l < n_mdps; ++l) free(pls[l].dist);
    free(pls);
    return 0;
}

/* ====================================================================================================
 * potential_function (mdp.py:2920-3238): phi(s) for potential-based reward shaping.
 *
 * Two pieces of the host language's runtime take part in the reference's result and are restated here:
 *  - `gamma ** k` is C `pow` (CPython float_pow / numpy npy_pow both call libm);
 *  - `get_partially_full_pots` (mdp.py:1882-1890) returns `list(set().union(...))` of position tuples, so the
 *    order of partially full pots — which breaks ties of the stable sort at mdp.py:3013-3024 and therefore decides
 *    which pot the greedy matching serves first — is CPython's set iteration order.  py_set_order restates
 *    CPython 3.10's tuple hash (Objects/tupleobject.c, xxHash-style) and open-addressing set (Objects/setobject.c:
 *    set_add_entry, set_table_resize, set_insert_clean); tests/test_oracle_golden.py checks it against the running
 *    interpreter's own `list(set().union(a, b))`.
 * ==================================================================================================== */
#include <math.h>

typedef struct PotentialParams { /* potential_params, mdp.py:2972-2982 */
    double gamma, tomato_value, onion_value;
    int32_t max_delivery_steps, max_pickup_steps, pot_onion_steps, pot_tomato_steps;
} PotentialParams;

static uint64_t py_tuple2_hash(uint64_t x, uint64_t y) { /* tuplehash of a 2-tuple of small non-negative ints */
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5, lane[2] = {x, y};
    for (int i = 0; i < 2; ++i) {
        acc += lane[i] * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 2 ^ (P5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796ULL;
    return acc;
}

typedef struct PySetEntry { int used; uint64_t hash; int value; } PySetEntry;

static void py_set_insert_clean(PySetEntry* table, size_t mask, uint64_t hash, int value) {
    size_t perturb = hash, i = (size_t)hash & mask;
    for (;;) {
        PySetEntry* e = &table[i];
        if (!e->used) { e->used = 1; e->hash = hash; e->value = value; return; }
        if (i + 9 <= mask) {
            for (int j = 0; j < 9; ++j) {
                ++e;
                if (!e->used) { e->used = 1; e->hash = hash; e->value = value; return; }
            }
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
}

/* order in which `list(set().union(*lists))` yields n distinct position tuples inserted in the given order */
static void py_set_order(const OracleMdp* m, const int* cells, int n, int* out) {
    PySetEntry tab_a[64], tab_b[64];
    PySetEntry* table = tab_a;
    size_t mask = 7, fill = 0;
    memset(tab_a, 0, sizeof(tab_a));
    for (int k = 0; k < n; ++k) {
        uint64_t hash = py_tuple2_hash((uint64_t)(cells[k] % m->width), (uint64_t)(cells[k] / m->width));
        size_t perturb = hash, i = (size_t)hash & mask;
        PySetEntry* slot = NULL;
        while (!slot) { /* set_add_entry; keys are distinct, so only the free-slot search remains */
            PySetEntry* e = &table[i];
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (!e->used) { slot = e; break; }
                ++e;
            } while (probes--);
            if (slot) break;
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
        slot->used = 1; slot->hash = hash; slot->value = cells[k];
        ++fill;
        if (fill * 5 >= mask * 3) { /* set_table_resize(so, used * 4) */
            size_t newsize = 8;
            while (newsize <= fill * 4) newsize <<= 1;
            PySetEntry* nt = (table == tab_a) ? tab_b : tab_a;
            memset(nt, 0, sizeof(tab_a));
            for (size_t s = 0; s <= mask; ++s)
                if (table[s].used) py_set_insert_clean(nt, newsize - 1, table[s].hash, table[s].value);
            table = nt;
            mask = newsize - 1;
        }
    }
    int k = 0;
    for (size_t s = 0; s <= mask; ++s)
        if (table[s].used) out[k++] = table[s].value;
}

/* get_recipe_value(..., discounted=True, base_recipe), mdp.py:1603-1628 */
static double discounted_recipe_value(const OracleMdp* m, int n_o, int n_t, int base_o, int base_t, const PotentialParams* pp) {
    int n_onions = n_o - base_o, n_tomatoes = n_t - base_t;
    return pow(pp->gamma, recipe_time(m, n_o, n_t)) * pow(pp->gamma, (double)(pp->pot_onion_steps * n_onions)) *
           pow(pp->gamma, (double)(pp->pot_tomato_steps * n_tomatoes)) * get_recipe_value(m, n_o, n_t);
}

/* _get_optimal_possible_recipe, mdp.py:1976-2016.  has_start == 0 is the reference's recipe=None. */
static void optimal_possible_recipe(const OracleMdp* m, int has_start, int s_o, int s_t, const PotentialParams* pp,
                                    int* best_o, int* best_t, double* best_value) {
    int visited[5][5] = {{0}}, stack[64][2], sp = 0;
    *best_o = has_start ? s_o : -1;
    *best_t = has_start ? s_t : -1;
    *best_value = 0.0;
    if (!has_start) { /* for ingredient in Recipe.ALL_INGREDIENTS = [onion, tomato] */
        stack[sp][0] = 1; stack[sp++][1] = 0;
        stack[sp][0] = 0; stack[sp++][1] = 1;
    } else { stack[sp][0] = s_o; stack[sp++][1] = s_t; }
    while (sp) {
        --sp;
        int o = stack[sp][0], t = stack[sp][1];
        if (visited[o][t]) continue;
        visited[o][t] = 1;
        double v = discounted_recipe_value(m, o, t, has_start ? s_o : 0, has_start ? s_t : 0, pp);
        if (v > *best_value) { *best_value = v; *best_o = o; *best_t = t; }
        if (o + t < m->max_num_ingredients) { /* Recipe.neighbors, mdp.py:193-205 */
            if (!visited[o + 1][t]) { stack[sp][0] = o + 1; stack[sp++][1] = t; }
            if (!visited[o][t + 1]) { stack[sp][0] = o; stack[sp++][1] = t + 1; }
        }
    }
}

/* mp.min_cost_to_feature(player.pos_and_or, cells) as a double (np.inf when no goal is reachable) */
static double cost_to(const OracleMdp* m, const Planner* pl, const Player* p, const int* cells, int n) {
    int cost = 0;
    int best = min_cost_to_feature(m, pl, cell_of(m, p->x, p->y), p->o, cells, n, &cost);
    return best < 0 ? INFINITY : (double)cost;
}

static double dmax(double a, double b) { return a >= b ? a : b; } /* Python max(a, b): b only if b > a */
static double dmin(double a, double b) { return b < a ? b : a; }

static double potential_one(const OracleMdp* m, const Planner* pl, const State* st, const PotentialParams* pp) {
    const double gamma = pp->gamma;
    int ncells = m->width * m->height;
    int pot_cells[MAX_CELLS], n_pots = 0, serve_cells[MAX_CELLS], n_serve = 0;
    for (int c = 0; c < ncells; ++c) {
        if (m->terrain[c] == 'P') pot_cells[n_pots++] = c;
        if (m->terrain[c] == 'S') serve_cells[n_serve++] = c;
    }
    PotStates ps;
    get_pot_states(m, st, &ps);

    /* steady state: geometric sum of optimal soups, mdp.py:2985-3001 */
    int opt_o, opt_t;
    double disc_value;
    optimal_possible_recipe(m, 0, 0, 0, pp, &opt_o, &opt_t, &disc_value);
    double opt_value = get_recipe_value(m, opt_o, opt_t);
    double discount = disc_value / opt_value;
    double potential = (discount / (1 - discount)) * opt_value;

    /* idle soups: full-but-not-cooking, then partially full (set order), stable-sorted by best completion value */
    int idle[MAX_POTS], n_idle = 0;
    for (int k = 0; k < n_pots; ++k) if (ps.cls[k] == m->max_num_ingredients) idle[n_idle++] = pot_cells[k];
    {
        int part[MAX_POTS], n_part = 0, ordered[MAX_POTS];
        for (int items = 1; items < m->max_num_ingredients; ++items)
            for (int k = 0; k < n_pots; ++k) if (ps.cls[k] == items) part[n_part++] = pot_cells[k];
        py_set_order(m, part, n_part, ordered);
        for (int k = 0; k < n_part; ++k) idle[n_idle++] = ordered[k];
    }
    double idle_key[MAX_POTS];
    for (int k = 0; k < n_idle; ++k) {
        int n_o, n_t, bo, bt;
        count_ing(&st->objects[idle[k]], &n_o, &n_t);
        optimal_possible_recipe(m, 1, n_o, n_t, pp, &bo, &bt, &idle_key[k]);
    }
    for (int a = 1; a < n_idle; ++a) { /* sorted(..., reverse=True) is stable: insertion sort, strict compare */
        int c = idle[a];
        double key = idle_key[a];
        int b = a - 1;
        while (b >= 0 && idle_key[b] < key) { idle[b + 1] = idle[b]; idle_key[b + 1] = idle_key[b]; --b; }
        idle[b + 1] = c;
        idle_key[b + 1] = key;
    }

    /* non-idle soups (cooking then ready, pot order) and their default values, mdp.py:3026-3046 */
    int non_idle[MAX_POTS], n_non_idle = 0;
    double non_idle_val[MAX_POTS];
    for (int k = 0; k < n_pots; ++k) if (ps.cls[k] == POT_COOKING) non_idle[n_non_idle++] = pot_cells[k];
    for (int k = 0; k < n_pots; ++k) if (ps.cls[k] == POT_READY) non_idle[n_non_idle++] = pot_cells[k];
    for (int k = 0; k < n_non_idle; ++k) {
        const Obj* soup = &st->objects[non_idle[k]];
        int n_o, n_t;
        count_ing(soup, &n_o, &n_t);
        double remaining = soup_cook_time(m, soup) - soup->tick;
        non_idle_val[k] = pow(gamma, pp->max_delivery_steps + dmax(pp->max_pickup_steps, remaining)) *
                          dmax(get_recipe_value(m, n_o, n_t), 1);
    }

    int holding_onion[2] = {0, 0}, holding_tomato[2] = {0, 0};
    for (int i = 0; i < m->n_players; ++i) {
        holding_onion[i] = st->players[i].held.name == NAME_ONION;
        holding_tomato[i] = st->players[i].held.name == NAME_TOMATO;
    }

    /* step 4: players holding soups, mdp.py:3078-3090 */
    for (int i = 0; i < m->n_players; ++i) {
        const Player* p = &st->players[i];
        if (p->held.name != NAME_SOUP) continue;
        int n_o, n_t;
        count_ing(&p->held, &n_o, &n_t);
        double delivery_dist = cost_to(m, pl, p, serve_cells, n_serve);
        potential += pow(gamma, dmin(delivery_dist, pp->max_delivery_steps)) * dmax(get_recipe_value(m, n_o, n_t), 1);
    }

    /* step 3: players holding dishes, mdp.py:3092-3133 */
    for (int i = 0; i < m->n_players; ++i) {
        const Player* p = &st->players[i];
        if (p->held.name != NAME_DISH) continue;
        int best = -1;
        double best_value = 0;
        for (int k = 0; k < n_non_idle; ++k) {
            const Obj* soup = &st->objects[non_idle[k]];
            int n_o, n_t;
            count_ing(soup, &n_o, &n_t);
            double pickup_dist = cost_to(m, pl, p, &non_idle[k], 1);
            double is_useful = pickup_dist < INFINITY ? 1.0 : 0.0;
            double pickup_soup_value = pow(gamma, pp->max_delivery_steps) * dmax(get_recipe_value(m, n_o, n_t), 1);
            double remaining = soup_cook_time(m, soup) - soup->tick;
            double disc = pow(gamma, dmax(remaining, dmin(pickup_dist, pp->max_pickup_steps)));
            double pickup_value = disc * pickup_soup_value * is_useful;
            if (pickup_dist < INFINITY && pickup_value > best_value) { best = k; best_value = pickup_value; }
        }
        if (best >= 0) non_idle_val[best] = dmax(non_idle_val[best], best_value);
    }
    for (int k = 0; k < n_non_idle; ++k) potential += non_idle_val[k];

    /* step 2: idle soups in decreasing value, mdp.py:3135-3211 */
    for (int k = 0; k < n_idle; ++k) {
        const Obj* soup = &st->objects[idle[k]];
        int n_o, n_t, bo, bt;
        double unused;
        count_ing(soup, &n_o, &n_t);
        optimal_possible_recipe(m, 1, n_o, n_t, pp, &bo, &bt, &unused);
        int missing[3], n_missing = 0; /* sorted ingredient tuple of the optimal recipe minus the soup's: onions first */
        for (int j = 0; j < bo - n_o; ++j) missing[n_missing++] = NAME_ONION;
        for (int j = 0; j < bt - n_t; ++j) missing[n_missing++] = NAME_TOMATO;
        double disc = pow(gamma, dmax(pp->max_pickup_steps, recipe_time(m, bo, bt)) + pp->max_delivery_steps);
        for (int j = 0; j < n_missing; ++j) {
            int* pertinent = missing[j] == NAME_TOMATO ? holding_tomato : holding_onion;
            double dist = INFINITY;
            int closest = -1;
            for (int i = 0; i < m->n_players; ++i) {
                if (!pertinent[i]) continue;
                double cur = cost_to(m, pl, &st->players[i], &idle[k], 1);
                if (cur < dist) { dist = cur; closest = i; }
            }
            disc *= pow(gamma, dmin(dist, missing[j] == NAME_TOMATO ? pp->pot_tomato_steps : pp->pot_onion_steps));
            if (closest >= 0) pertinent[closest] = 0;
        }
        if (n_missing) disc *= gamma;
        else {
            double cook_dist = INFINITY;
            for (int i = 0; i < m->n_players; ++i) {
                if (st->players[i].held.name != NAME_NONE) continue;
                double cur = cost_to(m, pl, &st->players[i], &idle[k], 1);
                if (cur < cook_dist) cook_dist = cur;
            }
            disc *= pow(gamma, dmin(cook_dist, pp->max_pickup_steps));
        }
        potential += disc * dmax(get_recipe_value(m, bo, bt), 1);
    }

    /* step 1: left-over ingredients go to empty pots, mdp.py:3213-3245 */
    int empty[MAX_POTS], n_empty = 0;
    for (int k = 0; k < n_pots; ++k) if (ps.cls[k] == POT_EMPTY) empty[n_empty++] = pot_cells[k];
    for (int pass = 0; pass < 2; ++pass) { /* tomatoes first, then onions */
        const int* holding = pass == 0 ? holding_tomato : holding_onion;
        for (int i = 0; i < m->n_players; ++i) {
            if (!holding[i]) continue;
            double dist = cost_to(m, pl, &st->players[i], empty, n_empty);
            double is_useful = dist < INFINITY ? 1.0 : 0.0;
            double steps = dmin(pass == 0 ? pp->pot_tomato_steps : pp->pot_onion_steps, dist) + pp->max_pickup_steps +
                           pp->max_delivery_steps;
            double disc = pow(gamma, steps) * is_useful;
            potential += disc * (pass == 0 ? pp->tomato_value : pp->onion_value);
        }
    }
    return potential;
}

/* phi: [n_envs] doubles; params: [n_mdps] */
int oracle_potential(const OracleMdp* mdps, int n_mdps, const uint16_t* layout_id, const PotentialParams* params,
                     const uint8_t* state, double* phi, int64_t n_envs) {
    Planner* pls = (Planner*)calloc((size_t)n_mdps, sizeof(Planner));
    for (int l = 0; l < n_mdps; ++l) planner_build(&mdps[l], NULL, &pls[l]);
    for (int64_t e = 0; e < n_envs; ++e) {
        int l = layout_id ? layout_id[e] : 0;
        State s;
        unpack_state(&mdps[l], state, n_envs, e, &s);
        phi[e] = potential_one(&mdps[l], &pls[l], &s, &params[l]);
    }
    for (int l = 0; l < n_mdps; ++l) free(pls[l].dist);
    free(pls);
    return 0;
}

/* test hook: CPython set iteration order of n distinct (x, y) cells inserted in order */
void oracle_py_set_order(int width, const int32_t* cells, int n, int32_t* out) {
    OracleMdp m;
    memset(&m, 0, sizeof(m));
    m.width = width;
    int in[MAX_POTS * 4] = {0}, o[MAX_POTS * 4] = {0}; /* n <= 18: the tables above hold 64 slots */
    for (int i = 0; i < n; ++i) in[i] = cells[i];
    py_set_order(&m, in, n, o);
    for (int i = 0; i < n; ++i) out[i] = o[i];
}

/* ====================================================================================================
 * Randomized start states: the start_state_fn of get_random_start_state_fn (mdp.py:1307-1369), with the draws
 * taken from the counter-based stream documented at oc_reset_random (include/oc_amd.h) instead of numpy's global
 * generator.  Structured like the reference: choose the joint position, build the standard start state, then
 * randomize pots and held objects.
 * ==================================================================================================== */
static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

static void random_soup(Obj* o, uint32_t word_n, uint32_t word_m, int tick) {
    int n = 1 + (int)mulhi32(word_n, 3u);          /* np.random.randint(low=1, high=4) */
    int m = (int)mulhi32(word_m, (uint32_t)(4 - n)); /* np.random.randint(low=0, high=4 - n) */
    o->name = NAME_SOUP;
    o->n_ing = n + m;
    for (int i = 0; i < n; ++i) o->ing[i] = NAME_ONION; /* SoupState.get_soup: onions + tomatoes, mdp.py:683-689 */
    for (int i = 0; i < m; ++i) o->ing[n + i] = NAME_TOMATO;
    o->tick = tick;
}

static void random_start_state(const OracleMdp* m, State* s, uint64_t seed, uint64_t g, uint32_t epoch,
                               int random_start_pos, uint64_t thresh) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x52535421u}, ctr[4], r[4];
    int ncells = m->width * m->height;
    standard_start_state(m, s);
    if (random_start_pos) {
        /* valid_positions[np.random.choice(len(valid_positions))], get_valid_joint_player_positions (mdp.py:1736-1747) */
        int floor[MAX_CELLS], n_floor = 0;
        for (int c = 0; c < ncells; ++c) if (m->terrain[c] == ' ') floor[n_floor++] = c;
        int joint[MAX_CELLS * MAX_CELLS][2], n_joint = 0;
        for (int a = 0; a < n_floor; ++a) {
            if (m->n_players == 1) { joint[n_joint][0] = floor[a]; joint[n_joint++][1] = -1; continue; }
            for (int b = 0; b < n_floor; ++b)
                if (a != b) { joint[n_joint][0] = floor[a]; joint[n_joint++][1] = floor[b]; }
        }
        ctr[0] = epoch; ctr[1] = (uint32_t)g; ctr[2] = (uint32_t)(g >> 32); ctr[3] = 0;
        oracle_philox4x32_10(ctr, key, r);
        int idx = (int)mulhi32(r[0], (uint32_t)n_joint);
        for (int i = 0; i < m->n_players; ++i) {
            s->players[i].x = joint[idx][i] % m->width;
            s->players[i].y = joint[idx][i] / m->width;
        }
    }
    if (thresh == 0) return; /* rnd_obj_prob_thresh == 0 */
    int k = 0;
    for (int c = 0; c < ncells; ++c) { /* pots in get_pot_locations order */
        if (m->terrain[c] != 'P') continue;
        ctr[0] = epoch; ctr[1] = (uint32_t)g; ctr[2] = (uint32_t)(g >> 32); ctr[3] = 3u + (uint32_t)k++;
        oracle_philox4x32_10(ctr, key, r);
        if ((uint64_t)r[0] < thresh) random_soup(&s->objects[c], r[1], r[2], (uint64_t)r[3] < thresh ? 0 : -1);
    }
    for (int i = 0; i < m->n_players; ++i) {
        ctr[0] = epoch; ctr[1] = (uint32_t)g; ctr[2] = (uint32_t)(g >> 32); ctr[3] = 1u + (uint32_t)i;
        oracle_philox4x32_10(ctr, key, r);
        if ((uint64_t)r[0] >= thresh) continue;
        Obj* h = &s->players[i].held;
        if (r[1] < 858993459u) { h->name = NAME_DISH; h->n_ing = 0; h->tick = -1; }        /* p = [0.2, 0.6, 0.2] */
        else if (r[1] < 3435973836u) { h->name = NAME_ONION; h->n_ing = 0; h->tick = -1; }
        else { /* finished=True -> auto_finish: tick = cook time (mdp.py:576-580) */
            random_soup(h, r[2], r[3], 0);
            h->tick = (int)soup_cook_time(m, h);
        }
    }
}

int oracle_reset_random(const OracleMdp* mdps, int n_mdps, const uint16_t* layout_id, uint8_t* state, const uint8_t* mask,
                        int64_t n_envs, uint64_t seed, int64_t env_offset, uint32_t epoch, int random_start_pos,
                        double rnd_obj_prob_thresh) {
    uint64_t thresh = (uint64_t)(rnd_obj_prob_thresh * 4294967296.0);
    for (int64_t e = 0; e < n_envs; ++e) {
        if (mask && !mask[e]) continue;
        const OracleMdp* m = &mdps[layout_id ? layout_id[e] : 0];
        State s;
        random_start_state(m, &s, seed, (uint64_t)(env_offset + e), epoch, random_start_pos, thresh);
        pack_state(m, &s, state, n_envs, e);
    }
    (void)n_mdps;
    return 0;
}
