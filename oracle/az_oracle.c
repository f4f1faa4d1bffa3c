/*
 * TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference's self-play
 * hot path, in plain C.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (the HIP engine in
 * alphazero-openspiel_amd/csrc) never does.
 *
 * Parity status: PINNED for tree search / agent / rollout loop — checked bit for
 * bit against fixtures produced by running the real reference here
 * (oracle/gen_golden.py -> tests/golden/{mcts_trace,selfplay,remove_illegal}.json,
 * tests/test_oracle_golden.py).  Game RULES are restated from the public game
 * definitions: OpenSpiel (pyspiel) is a third-party dependency absent from
 * /root/reference, no version is pinned by the reference and it ships no rule
 * tests => "parity unpinned" for apply_action/legal_actions/returns, except the
 * observation-plane order and breakthrough action codec, which the shipped
 * checkpoints pin (tests/test_checkpoint_pins.py).
 *
 * Each function cites the reference lines it follows.  The structure mirrors the
 * reference deliberately (heap nodes with parent pointers, per-node child
 * lists in insertion order, recursive backup, cell-array boards) so that it is an
 * independent implementation from the HIP engine's SoA pools and bitboards.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: the reference's arithmetic is Python float (IEEE
 * double, one rounding per operation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_CELLS 64
#define ORC_MAX_HIST 512
#define ORC_GAME_CONNECT_FOUR 0
#define ORC_GAME_BREAKTHROUGH 1
#define ORC_TERMINAL_PLAYER (-4)

/* ------------------------------------------------------------------ games */
typedef struct {
    int32_t game, rows, cols;
    int8_t cell[ORC_MAX_CELLS]; /* observation plane index of the cell's content */
    int32_t pieces[2];          /* breakthrough only */
    int32_t nhist;
    int32_t hist[ORC_MAX_HIST];
    int32_t terminal;
    double ret0; /* returns()[0] */
} orc_state;

int orc_state_size(void) { return (int)sizeof(orc_state); }

int orc_num_actions(int game, int rows, int cols) {
    return game == ORC_GAME_CONNECT_FOUR ? cols : rows * cols * 6 * 2;
}

void orc_state_init(orc_state *s, int game, int rows, int cols) {
    memset(s, 0, sizeof *s);
    s->game = game;
    s->rows = rows;
    s->cols = cols;
    if (game == ORC_GAME_CONNECT_FOUR) {
        /* plane 0 = empty, 1 = player-1 'o', 2 = player-0 'x' */
        for (int i = 0; i < rows * cols; i++) s->cell[i] = 0;
    } else {
        /* plane 0 = black (player 0, rows 0..1), 1 = white, 2 = empty */
        for (int i = 0; i < rows * cols; i++) s->cell[i] = 2;
        for (int c = 0; c < cols; c++) {
            s->cell[0 * cols + c] = 0;
            s->cell[1 * cols + c] = 0;
            s->cell[(rows - 2) * cols + c] = 1;
            s->cell[(rows - 1) * cols + c] = 1;
        }
        s->pieces[0] = s->pieces[1] = 2 * cols;
    }
}

int orc_is_terminal(const orc_state *s) { return s->terminal; }
int orc_current_player(const orc_state *s) { return s->terminal ? ORC_TERMINAL_PLAYER : (s->nhist & 1); }
double orc_player_return(const orc_state *s, int player) { return player == 0 ? s->ret0 : -s->ret0; }

static const int BT_DR[6] = {1, 1, 1, -1, -1, -1};
static const int BT_DC[6] = {-1, 0, 1, -1, 0, 1};

/* legal_actions(current_player) in ascending order (mcts.py:147,184; alphazerobot.py:72) */
int orc_legal_actions(const orc_state *s, int32_t *out) {
    int n = 0;
    if (s->terminal) return 0;
    int R = s->rows, C = s->cols;
    if (s->game == ORC_GAME_CONNECT_FOUR) {
        for (int c = 0; c < C; c++)
            if (s->cell[(R - 1) * C + c] == 0) out[n++] = c;
        return n;
    }
    int me = s->nhist & 1;
    int d0 = me == 0 ? 0 : 3;
    for (int r = 0; r < R; r++)
        for (int c = 0; c < C; c++) {
            if (s->cell[r * C + c] != me) continue;
            for (int d = d0; d < d0 + 3; d++) {
                int r2 = r + BT_DR[d], c2 = c + BT_DC[d];
                if (r2 < 0 || r2 >= R || c2 < 0 || c2 >= C) continue;
                int t = s->cell[r2 * C + c2];
                if (t == 2)
                    out[n++] = ((r * C + c) * 6 + d) * 2;
                else if (t == 1 - me && BT_DC[d] != 0)
                    out[n++] = ((r * C + c) * 6 + d) * 2 + 1;
            }
        }
    return n;
}

static int c4_line(const orc_state *s, int r, int c, int dr, int dc, int mark) {
    int n = 1;
    for (int sg = -1; sg <= 1; sg += 2) {
        int rr = r + sg * dr, cc = c + sg * dc;
        while (rr >= 0 && rr < s->rows && cc >= 0 && cc < s->cols && s->cell[rr * s->cols + cc] == mark) {
            n++;
            rr += sg * dr;
            cc += sg * dc;
        }
    }
    return n;
}

/* returns 0 ok, <0 illegal */
int orc_apply_action(orc_state *s, int action) {
    if (s->terminal || s->nhist >= ORC_MAX_HIST) return -1;
    int R = s->rows, C = s->cols, me = s->nhist & 1;
    if (s->game == ORC_GAME_CONNECT_FOUR) {
        if (action < 0 || action >= C) return -2;
        int mark = me == 0 ? 2 : 1, r = 0;
        while (r < R && s->cell[r * C + action] != 0) r++;
        if (r >= R) return -3;
        s->cell[r * C + action] = (int8_t)mark;
        s->hist[s->nhist++] = action;
        if (c4_line(s, r, action, 0, 1, mark) >= 4 || c4_line(s, r, action, 1, 0, mark) >= 4 ||
            c4_line(s, r, action, 1, 1, mark) >= 4 || c4_line(s, r, action, 1, -1, mark) >= 4) {
            s->terminal = 1;
            s->ret0 = me == 0 ? 1.0 : -1.0;
        } else if (s->nhist == R * C) {
            s->terminal = 1;
            s->ret0 = 0.0;
        }
        return 0;
    }
    if (action < 0 || action >= R * C * 12) return -2;
    int cap = action & 1, d = (action >> 1) % 6, cell = (action >> 1) / 6;
    int r = cell / C, c = cell % C, r2 = r + BT_DR[d], c2 = c + BT_DC[d];
    if (s->cell[cell] != me || r2 < 0 || r2 >= R || c2 < 0 || c2 >= C) return -3;
    if ((me == 0) != (d < 3)) return -3;
    int t = s->cell[r2 * C + c2];
    if (cap) {
        if (t != 1 - me || BT_DC[d] == 0) return -4;
        s->pieces[1 - me]--;
    } else if (t != 2)
        return -4;
    s->cell[cell] = 2;
    s->cell[r2 * C + c2] = (int8_t)me;
    s->hist[s->nhist++] = action;
    if ((me == 0 && r2 == R - 1) || (me == 1 && r2 == 0) || s->pieces[1 - me] == 0) {
        s->terminal = 1;
        s->ret0 = me == 0 ? 1.0 : -1.0;
    }
    return 0;
}

/* network.py:9-18 state_to_board: planes 0..2 = observation, plane 3 = current_player */
void orc_state_to_board(const orc_state *s, double *out) {
    int n = s->rows * s->cols;
    for (int i = 0; i < 4 * n; i++) out[i] = 0.0;
    for (int i = 0; i < n; i++) out[s->cell[i] * n + i] = 1.0;
    double p = (double)orc_current_player(s);
    for (int i = 0; i < n; i++) out[3 * n + i] = p;
}

/* ------------------------------------------------------------------ Node (mcts.py:10-89) */
typedef struct orc_node {
    struct orc_node *parent;
    int32_t n_children, cap_children;
    int32_t *actions;           /* dict keys, insertion order */
    struct orc_node **children; /* dict values */
    double P, Q;
    int64_t N;
    int use_puct;
} orc_node;

static int64_t g_nodes_alive = 0, g_nodes_total = 0;

static orc_node *node_new(orc_node *parent, double prior_p, int use_puct) { /* mcts.py:14-20 */
    orc_node *n = (orc_node *)calloc(1, sizeof *n);
    n->parent = parent;
    n->P = prior_p;
    n->Q = 0.0;
    n->N = 0;
    n->use_puct = use_puct;
    g_nodes_alive++;
    g_nodes_total++;
    return n;
}

static void node_free(orc_node *n) {
    if (!n) return;
    for (int i = 0; i < n->n_children; i++) node_free(n->children[i]);
    free(n->actions);
    free(n->children);
    free(n);
    g_nodes_alive--;
}

static int node_is_leaf(const orc_node *n) { return n->n_children == 0; } /* mcts.py:22-28 */

static orc_node *node_child(const orc_node *n, int action) {
    for (int i = 0; i < n->n_children; i++)
        if (n->actions[i] == action) return n->children[i];
    return NULL;
}

/* mcts.py:68-80 */
static double node_get_value(const orc_node *n, double c_puct) {
    if (n->use_puct) /* self.Q + c_puct * self.P * math.sqrt(self.parent.N) / (self.N+1) */
        return n->Q + c_puct * n->P * sqrt((double)n->parent->N) / (double)(n->N + 1);
    if (n->N == 0) return INFINITY;
    return n->Q + c_puct * n->P * sqrt(log((double)n->parent->N) / (double)n->N);
}

/* mcts.py:38-52: max() over the dict returns the FIRST maximal key in insertion order */
static orc_node *node_select(const orc_node *n, double c_puct, int *action) {
    int best = 0;
    double bv = node_get_value(n->children[0], c_puct);
    for (int i = 1; i < n->n_children; i++) {
        double v = node_get_value(n->children[i], c_puct);
        if (v > bv) {
            bv = v;
            best = i;
        }
    }
    *action = n->actions[best];
    return n->children[best];
}

/* mcts.py:54-66 */
static void node_expand(orc_node *n, const double *prior_ps, const int32_t *legal, int n_legal) {
    for (int i = 0; i < n_legal; i++) {
        orc_node *c = node_child(n, legal[i]);
        if (!c) {
            if (n->n_children == n->cap_children) {
                n->cap_children = n->cap_children ? 2 * n->cap_children : 8;
                n->actions = (int32_t *)realloc(n->actions, sizeof(int32_t) * n->cap_children);
                n->children = (orc_node **)realloc(n->children, sizeof(orc_node *) * n->cap_children);
            }
            n->actions[n->n_children] = legal[i];
            n->children[n->n_children++] = node_new(n, prior_ps[legal[i]], n->use_puct);
        } else
            c->P = prior_ps[legal[i]];
    }
}

/* mcts.py:82-89 */
static void node_update(orc_node *n, double value) {
    n->Q = ((double)n->N * n->Q + value) / (double)(n->N + 1);
    n->N += 1;
}
static void node_update_recursive(orc_node *n, double value) {
    if (n->parent) node_update_recursive(n->parent, -value);
    node_update(n, value);
}

/* ------------------------------------------------------------------ MCTS (mcts.py:92-203) */
typedef void (*orc_policy_fn)(void *user, const orc_state *s, const double *board, double *priors, double *value);

typedef struct {
    int32_t num_actions;
    double c_puct;
    int32_t n_playouts, use_dirichlet, use_puct;
    double dirichlet_ratio;
    orc_node *root;
    orc_policy_fn policy_fn;
    void *user;
    /* statistics for the roofline accounting */
    int64_t n_sims, n_evals, n_terminal_hits, sum_depth, sum_children_seen;
    /* scratch */
    double *board, *priors;
} orc_mcts;

orc_mcts *orc_mcts_new(int num_actions, double c_puct, int n_playouts, int use_dirichlet,
                       double dirichlet_ratio, int use_puct, orc_policy_fn fn, void *user) {
    orc_mcts *m = (orc_mcts *)calloc(1, sizeof *m);
    m->num_actions = num_actions;
    m->c_puct = c_puct;
    m->n_playouts = n_playouts;
    m->use_dirichlet = use_dirichlet;
    m->use_puct = use_puct;
    m->dirichlet_ratio = dirichlet_ratio;
    m->root = node_new(NULL, 0.0, 1); /* mcts.py:122 — always use_puct=True here */
    m->policy_fn = fn;
    m->user = user;
    m->board = (double *)malloc(sizeof(double) * 4 * ORC_MAX_CELLS);
    m->priors = (double *)malloc(sizeof(double) * num_actions);
    return m;
}

void orc_mcts_free(orc_mcts *m) {
    if (!m) return;
    node_free(m->root);
    free(m->board);
    free(m->priors);
    free(m);
}

static void call_policy(orc_mcts *m, const orc_state *s, double *value) {
    orc_state_to_board(s, m->board);
    m->policy_fn(m->user, s, m->board, m->priors, value);
    m->n_evals++;
}

/* mcts.py:126-153 — `state` is the caller's clone and is modified */
void orc_mcts_playout(orc_mcts *m, orc_state *state) {
    orc_node *node = m->root;
    int32_t legal[ORC_MAX_CELLS * 3];
    int current_player = orc_current_player(state);
    int depth = 0;
    while (!node_is_leaf(node) && !orc_is_terminal(state)) {
        int action;
        current_player = orc_current_player(state);
        m->sum_children_seen += node->n_children;
        node = node_select(node, m->c_puct, &action);
        orc_apply_action(state, action);
        depth++;
    }
    double leaf_value;
    if (!orc_is_terminal(state)) {
        call_policy(m, state, &leaf_value);
        int n = orc_legal_actions(state, legal);
        node_expand(node, m->priors, legal, n);
    } else {
        leaf_value = -orc_player_return(state, current_player);
        m->n_terminal_hits++;
    }
    node_update_recursive(node, -leaf_value);
    m->n_sims++;
    m->sum_depth += depth;
}

/* mcts.py:182-190; eta = the Dirichlet(0.3) draw of length n_legal */
void orc_mcts_expand_root_dirichlet(orc_mcts *m, const orc_state *state, const double *eta) {
    int32_t legal[ORC_MAX_CELLS * 3];
    double v;
    call_policy(m, state, &v);
    int n = orc_legal_actions(state, legal);
    for (int a = 0; a < m->num_actions; a++) m->priors[a] = (1.0 - m->dirichlet_ratio) * m->priors[a];
    for (int i = 0; i < n; i++) m->priors[legal[i]] = m->priors[legal[i]] + 0.25 * eta[i];
    node_expand(m->root, m->priors, legal, n);
}

/* mcts.py:155-162 */
void orc_mcts_visit_counts(const orc_mcts *m, double *pi) {
    int64_t tot = 0;
    for (int i = 0; i < m->root->n_children; i++) tot += m->root->children[i]->N;
    for (int a = 0; a < m->num_actions; a++) pi[a] = 0.0 / (double)tot; /* ZeroDivisionError analogue: NaN/0 */
    for (int i = 0; i < m->root->n_children; i++)
        pi[m->root->actions[i]] = (double)m->root->children[i]->N / (double)tot;
}

/* mcts.py:164-180 */
void orc_mcts_search(orc_mcts *m, const orc_state *state, const double *eta, double *pi) {
    if (m->use_dirichlet) orc_mcts_expand_root_dirichlet(m, state, eta);
    for (int i = 0; i < m->n_playouts; i++) {
        orc_state copy = *state;
        orc_mcts_playout(m, &copy);
    }
    orc_mcts_visit_counts(m, pi);
}

/* mcts.py:192-203 */
void orc_mcts_update_root(orc_mcts *m, int action) {
    if (node_is_leaf(m->root)) {
        node_free(m->root);
        m->root = node_new(NULL, 0.0, m->use_puct);
        return;
    }
    orc_node *old = m->root, *keep = NULL;
    for (int i = 0; i < old->n_children; i++)
        if (old->actions[i] == action) {
            keep = old->children[i];
            old->children[i] = NULL;
        }
    /* free siblings (the reference leaves them to the GC) */
    for (int i = 0; i < old->n_children; i++)
        if (old->children[i]) node_free(old->children[i]);
    old->n_children = 0;
    node_free(old);
    keep->parent = NULL;
    m->root = keep;
}

/* root read-back for tests: returns n_children, fills arrays (insertion = ascending action order) */
int orc_mcts_root_stats(const orc_mcts *m, int64_t *rootN, double *rootQ, int32_t *actions, int64_t *cN,
                        double *cQ, double *cP) {
    *rootN = m->root->N;
    *rootQ = m->root->Q;
    for (int i = 0; i < m->root->n_children; i++) {
        actions[i] = m->root->actions[i];
        cN[i] = m->root->children[i]->N;
        cQ[i] = m->root->children[i]->Q;
        cP[i] = m->root->children[i]->P;
    }
    return m->root->n_children;
}

void orc_mcts_counters(const orc_mcts *m, int64_t *out5) {
    out5[0] = m->n_sims;
    out5[1] = m->n_evals;
    out5[2] = m->n_terminal_hits;
    out5[3] = m->sum_depth;
    out5[4] = m->sum_children_seen;
}

/* ------------------------------------------------------------------ numpy arithmetic used on the path */
/* numpy's pairwise summation (np.sum of a contiguous float64 vector), as used by
 * remove_illegal_actions (alphazerobot.py:13-14). */
static double np_pairwise_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.0; /* numpy starts from a[0]; 0.0 + a[0] == a[0] for our non-negative inputs, -0.0 aside */
        if (n == 0) return 0.0;
        res = a[0];
        for (int i = 1; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

double orc_np_sum(const double *a, int n) { return np_pairwise_sum(a, n); }

/* numpy's scalar-exponent fast paths for ndarray ** float */
static double np_pow(double x, double e) {
    if (e == 1.0) return x;
    if (e == 2.0) return x * x;
    if (e == 0.5) return sqrt(x);
    if (e == -1.0) return 1.0 / x;
    return pow(x, e);
}

/* alphazerobot.py:7-18; probs is modified in place / replaced, length A */
void orc_remove_illegal_actions(double *probs, int A, const int32_t *legal, int n_legal) {
    char *ok = (char *)calloc(A, 1);
    for (int i = 0; i < n_legal; i++) ok[legal[i]] = 1;
    for (int a = 0; a < A; a++)
        if (!ok[a]) probs[a] = 0.0;
    double s = np_pairwise_sum(probs, A);
    if (s > 1e-6) {
        for (int a = 0; a < A; a++) probs[a] = probs[a] / s;
    } else {
        for (int a = 0; a < A; a++) probs[a] = 0.0;
        for (int i = 0; i < n_legal; i++) probs[legal[i]] = 1.0 / (double)n_legal;
    }
    free(ok);
}

/* np.random.choice(n, p=p) given its single uniform draw u */
static int np_choice(const double *p, int n, double u) {
    double *cdf = (double *)malloc(sizeof(double) * n);
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
        acc += p[i];
        cdf[i] = acc;
    }
    double last = cdf[n - 1];
    for (int i = 0; i < n; i++) cdf[i] /= last;
    /* searchsorted(side='right'): first i with cdf[i] > u */
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (cdf[mid] <= u)
            lo = mid + 1;
        else
            hi = mid;
    }
    free(cdf);
    return lo;
}

/* ------------------------------------------------------------------ internal RNG for un-injected runs */
typedef struct { uint64_t s; } orc_rng;
static uint64_t rng_next(orc_rng *r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rng_u01(orc_rng *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static double rng_normal(orc_rng *r) {
    double u1 = 1.0 - rng_u01(r), u2 = rng_u01(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static double rng_gamma(orc_rng *r, double alpha) { /* Marsaglia-Tsang, alpha<1 via boost */
    if (alpha < 1.0) {
        double u = 1.0 - rng_u01(r);
        return rng_gamma(r, alpha + 1.0) * pow(u, 1.0 / alpha);
    }
    double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = rng_normal(r), v = 1.0 + c * x;
        if (v <= 0) continue;
        v = v * v * v;
        double u = 1.0 - rng_u01(r);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v;
    }
}
static void rng_dirichlet(orc_rng *r, double alpha, int n, double *out) {
    double s = 0;
    for (int i = 0; i < n; i++) s += (out[i] = rng_gamma(r, alpha));
    double inv = 1.0 / s;
    for (int i = 0; i < n; i++) out[i] *= inv;
}

/* ------------------------------------------------------------------ AlphaZeroBot (alphazerobot.py:21-93) */
typedef struct {
    orc_mcts *mcts;
    int32_t num_actions, self_play, keep_search_tree, use_probabilistic_actions;
    int32_t num_probabilistic_actions; /* alphazerobot.py:36 */
    double temperature;
    /* ctor args kept to rebuild the tree when keep_search_tree is False */
    double c_puct, dirichlet_ratio;
    int32_t n_playouts, use_dirichlet, use_puct;
    orc_policy_fn fn;
    void *user;
} orc_bot;

orc_bot *orc_bot_new(int num_actions, int self_play, int keep_search_tree, double temperature, double c_puct,
                     int n_playouts, int use_dirichlet, double dirichlet_ratio, int use_puct,
                     orc_policy_fn fn, void *user) {
    orc_bot *b = (orc_bot *)calloc(1, sizeof *b);
    b->num_actions = num_actions;
    b->self_play = self_play;
    b->keep_search_tree = keep_search_tree;
    b->use_probabilistic_actions = self_play; /* alphazerobot.py:32 */
    b->num_probabilistic_actions = 1000;      /* alphazerobot.py:36 */
    b->temperature = temperature;
    b->c_puct = c_puct;
    b->dirichlet_ratio = dirichlet_ratio;
    b->n_playouts = n_playouts;
    b->use_dirichlet = use_dirichlet;
    b->use_puct = use_puct;
    b->fn = fn;
    b->user = user;
    b->mcts = orc_mcts_new(num_actions, c_puct, n_playouts, use_dirichlet, dirichlet_ratio, use_puct, fn, user);
    return b;
}
void orc_bot_free(orc_bot *b) {
    if (!b) return;
    orc_mcts_free(b->mcts);
    free(b);
}
orc_mcts *orc_bot_mcts(orc_bot *b) { return b->mcts; }

/* alphazerobot.py:42-93.  eta: Dirichlet draw for this move (n_legal), u: the uniform behind
 * np.random.choice.  policy_out: dense A un-tempered masked visit distribution. returns action */
int orc_bot_step(orc_bot *b, const orc_state *state, const double *eta, double u, double *policy_out) {
    int A = b->num_actions;
    if (b->keep_search_tree) {
        if (b->self_play) {
            if (state->nhist) orc_mcts_update_root(b->mcts, state->hist[state->nhist - 1]);
        } else if (state->nhist >= 2) {
            orc_mcts_update_root(b->mcts, state->hist[state->nhist - 2]);
            orc_mcts_update_root(b->mcts, state->hist[state->nhist - 1]);
        }
    } else {
        int64_t keep[5];
        orc_mcts_counters(b->mcts, keep);
        orc_mcts_free(b->mcts);
        b->mcts = orc_mcts_new(A, b->c_puct, b->n_playouts, b->use_dirichlet, b->dirichlet_ratio, b->use_puct,
                               b->fn, b->user);
        b->mcts->n_sims = keep[0];
        b->mcts->n_evals = keep[1];
        b->mcts->n_terminal_hits = keep[2];
        b->mcts->sum_depth = keep[3];
        b->mcts->sum_children_seen = keep[4];
    }
    double *nv = policy_out;
    orc_mcts_search(b->mcts, state, eta, nv);
    int32_t legal[ORC_MAX_CELLS * 3];
    int n_legal = orc_legal_actions(state, legal);
    orc_remove_illegal_actions(nv, A, legal, n_legal);
    /* action_probabilities = nv**(1/T) / sum(nv**(1/T))   (builtin sum: left to right) */
    double *ap = (double *)malloc(sizeof(double) * A);
    double e = 1.0 / b->temperature, tot = 0.0;
    for (int a = 0; a < A; a++) {
        ap[a] = np_pow(nv[a], e);
        tot = (a == 0) ? (0 + ap[a]) : tot + ap[a];
    }
    for (int a = 0; a < A; a++) ap[a] = ap[a] / tot;
    int action;
    if (b->use_probabilistic_actions && state->nhist < b->num_probabilistic_actions)
        action = np_choice(ap, A, u);
    else { /* np.argmax: first maximum */
        action = 0;
        for (int a = 1; a < A; a++)
            if (ap[a] > ap[action]) action = a;
    }
    free(ap);
    return action;
}

/* ------------------------------------------------------------------ play_game_self (game_utils.py:148-206) */
#define ORC_BACKUP_ON_POLICY 0
#define ORC_BACKUP_SOFT_Z 1
#define ORC_BACKUP_A0C 2
#define ORC_BACKUP_OFF_POLICY 3

typedef struct {
    int32_t game, rows, cols;
    int32_t n_playouts, use_dirichlet, use_puct, keep_search_tree, backup;
    double c_puct, dirichlet_ratio, temperature;
    uint64_t seed; /* used only when etas/us are NULL */
    int32_t max_moves; /* >0: stop after this many moves (bounded timing sample); 0 = play to the end */
    int32_t num_probabilistic_actions; /* alphazerobot.py:36; 0 = the default 1000, < 0 = never sample */
} orc_selfplay_cfg;

/* off-policy / A0GB target (game_utils.py:182-194) */
static double a0gb_value(const orc_node *root) {
    const orc_node *node = root;
    double value = 0.0, value_mult = 1.0;
    while (!node_is_leaf(node)) {
        value = node->Q;
        int best = 0;
        double bv = 0;
        for (int i = 0; i < node->n_children; i++) {
            const orc_node *c = node->children[i];
            double v = c->N > 0 ? (double)c->N + c->P : -99.0;
            if (i == 0 || v > bv) {
                bv = v;
                best = i;
            }
        }
        node = node->children[best];
        value_mult *= -1.0;
    }
    if (node->N > 0) {
        value = node->Q;
        value_mult *= -1.0;
    }
    return value * value_mult;
}

/*
 * Plays one game.  Outputs (caller-allocated, max_plies rows):
 *   boards [max_plies][4*R*C], pis [max_plies][A], values [max_plies], actions [max_plies]
 *   root_cN [max_plies][maxc] visit counts of the root's children before the move (maxc = 3*cells cap given)
 * etas: [max_plies][eta_stride] injected Dirichlet draws (may be NULL -> internal RNG)
 * us:   [max_plies] injected choice uniforms          (may be NULL -> internal RNG)
 * returns number of plies, or <0 on overflow.  counters5 as orc_mcts_counters.
 */
int orc_play_game_self(const orc_selfplay_cfg *cfg, orc_policy_fn fn, void *user, const double *etas,
                       int eta_stride, const double *us, int max_plies, double *boards, double *pis,
                       double *values, int32_t *actions, int64_t *root_cN, int cn_stride, double *ret0,
                       int64_t *counters5) {
    orc_state st;
    orc_state_init(&st, cfg->game, cfg->rows, cfg->cols);
    int A = orc_num_actions(cfg->game, cfg->rows, cfg->cols);
    int ncell4 = 4 * cfg->rows * cfg->cols;
    orc_bot *bot = orc_bot_new(A, 1, cfg->keep_search_tree, cfg->temperature, cfg->c_puct, cfg->n_playouts,
                               cfg->use_dirichlet, cfg->dirichlet_ratio, cfg->use_puct, fn, user);
    if (cfg->num_probabilistic_actions) bot->num_probabilistic_actions = cfg->num_probabilistic_actions > 0 ? cfg->num_probabilistic_actions : 0;
    orc_rng rng = {cfg->seed};
    double eta_buf[ORC_MAX_CELLS * 3];
    int32_t legal[ORC_MAX_CELLS * 3];
    int ply = 0;
    while (!orc_is_terminal(&st) && !(cfg->max_moves > 0 && ply >= cfg->max_moves)) {
        if (ply >= max_plies) {
            orc_bot_free(bot);
            return -1;
        }
        const double *eta = NULL;
        if (cfg->use_dirichlet) {
            if (etas)
                eta = etas + (size_t)ply * eta_stride;
            else {
                int n = orc_legal_actions(&st, legal);
                rng_dirichlet(&rng, 0.3, n, eta_buf);
                eta = eta_buf;
            }
        }
        double u = us ? us[ply] : rng_u01(&rng);
        double *pi = pis + (size_t)ply * A;
        int action = orc_bot_step(bot, &st, eta, u, pi);
        /* policy_list: pi over legal actions, 0 elsewhere — pi already is that (game_utils.py:160-164) */
        orc_state_to_board(&st, boards + (size_t)ply * ncell4);
        const orc_node *root = bot->mcts->root;
        if (root_cN)
            for (int i = 0; i < cn_stride; i++)
                root_cN[(size_t)ply * cn_stride + i] = i < root->n_children ? root->children[i]->N : -1;
        double val = 0.0;
        switch (cfg->backup) {
        case ORC_BACKUP_SOFT_Z: /* game_utils.py:172-174 */
            val = -root->Q;
            break;
        case ORC_BACKUP_A0C: { /* game_utils.py:177-179 */
            val = -INFINITY;
            for (int i = 0; i < root->n_children; i++) {
                double q = root->children[i]->N > 0 ? root->children[i]->Q : -99.0;
                if (q > val) val = q;
            }
            break;
        }
        case ORC_BACKUP_OFF_POLICY:
            val = a0gb_value(root);
            break;
        default:
            break;
        }
        values[ply] = val;
        actions[ply] = action;
        orc_apply_action(&st, action); /* game_utils.py:197 */
        ply++;
    }
    if (cfg->backup == ORC_BACKUP_ON_POLICY) { /* game_utils.py:200-204 */
        double reward = st.ret0;
        for (int i = 0; i < ply; i++) {
            values[i] = reward;
            reward *= -1;
        }
    }
    *ret0 = st.ret0;
    if (counters5) orc_mcts_counters(bot->mcts, counters5);
    orc_bot_free(bot);
    return ply;
}


/* ================================================================== evaluation arena (game_utils.py:16-145)
 * Agent (network driven) against an opponent bot.  The opponents come from OpenSpiel, which is absent from the reference
 * tree and unpinned (see the header): `pyspiel.make_uniform_random_bot` and `open_spiel.python.algorithms.mcts.MCTSBot(game,
 * player, uct_c, max_search_nodes, RandomRolloutEvaluator(1))` are restated from their published behaviour - "parity
 * unpinned" for them; what IS pinned here is that the HIP arena (csrc/az_engine.hip: az_opponent_kernel, move_step) computes
 * exactly this, given the same counter-based random stream (Philox4x32-10, Salmon et al. 2011, keyed by seed / game id /
 * ply / purpose exactly as the device keys it). */
typedef struct {
    uint32_t k0, k1, c0, c1, c2, c3, out[4];
    int have;
} orc_philox;
static void philox_init(orc_philox *r, uint64_t seed, uint32_t gid, uint32_t ply, uint32_t purpose, uint32_t idx) {
    r->k0 = (uint32_t)seed;
    r->k1 = (uint32_t)(seed >> 32);
    r->c0 = 0;
    r->c1 = idx;
    r->c2 = (ply << 8) | purpose;
    r->c3 = gid;
    r->have = 0;
}
static void philox_block(orc_philox *r) {
    uint32_t c0 = r->c0, c1 = r->c1, c2 = r->c2, c3 = r->c3, k0 = r->k0, k1 = r->k1;
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r->out[0] = c0; r->out[1] = c1; r->out[2] = c2; r->out[3] = c3;
    r->c0++;
    r->have = 2;
}
static double philox_u01(orc_philox *r) { /* [0,1), 53 bits; two per block, upper pair first */
    if (!r->have) philox_block(r);
    r->have--;
    uint64_t x = ((uint64_t)r->out[2 * r->have] << 32) | r->out[2 * r->have + 1];
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* one uniformly random legal action */
static int random_legal_action(const orc_state *s, orc_philox *r) {
    int32_t legal[ORC_MAX_CELLS * 3];
    int n = orc_legal_actions(s, legal);
    int k = (int)(philox_u01(r) * (double)n);
    return legal[k < n ? k : n - 1];
}

typedef struct uct_node {
    int64_t explore_count;
    double total_reward; /* seen by the player who moved into this node */
    int action, n_children;
    int solved;          /* MCTS-Solver: the game-theoretic value below this node is known ... */
    double outcome0;     /* ... and this is player 0's return under it */
    struct uct_node *children; /* created the first time a visited node is descended through, in SHUFFLED legal-action order */
} uct_node;
static void uct_free(uct_node *n) {
    for (int i = 0; i < n->n_children; i++) uct_free(&n->children[i]);
    free(n->children);
}
/* MCTSBot.step as OpenSpiel publishes it (open_spiel/python/algorithms/mcts.py; third party, absent here, restated: "parity
 * unpinned"), with its defaults solve=True and child_selection_fn=SearchNode.uct_value:
 *   tree policy   descend while the node has been visited; a visited node without children gets them on the way through, in
 *                 an order SHUFFLED by the bot's random state (here: Fisher-Yates on the Philox stream);
 *                 child value = its proven outcome for the player who moved into it, else +inf when unvisited, else
 *                 total_reward / explore_count + uct_c * sqrt(log(parent explore_count) / explore_count); first maximum;
 *   evaluation    a terminal node gets its returns as outcome (solved); any other leaf ONE uniformly random rollout;
 *   backup        reward and count along the path; while "solved": a node whose children are all solved, or one of whose
 *                 children is a proven win for the player to move, takes the outcome of its best child; else solved ends;
 *   stop          after max_simulations, or as soon as the root is solved;
 *   move          the child with the largest (proven outcome for the mover or 0, explore_count, total_reward); first maximum. */
static int uct_bot_action(const orc_state *root_state, int n_sims, double uct_c, orc_philox *r) {
    uct_node root;
    memset(&root, 0, sizeof root);
    int root_player = root_state->nhist & 1;
    uct_node *path[ORC_MAX_HIST + 1];
    for (int sim = 0; sim < n_sims; sim++) {
        orc_state s = *root_state;
        uct_node *node = &root;
        int depth = 0;
        path[0] = node;
        while (!s.terminal && node->explore_count > 0) {
            const int to_move = (root_player + depth) & 1; /* the player who moves INTO the children */
            if (!node->children) {
                int32_t legal[ORC_MAX_CELLS * 3];
                int n = orc_legal_actions(&s, legal);
                for (int i = n - 1; i >= 1; i--) { /* random_state.shuffle */
                    int j = (int)(philox_u01(r) * (double)(i + 1));
                    if (j > i) j = i;
                    int32_t t = legal[i];
                    legal[i] = legal[j];
                    legal[j] = t;
                }
                node->children = (uct_node *)calloc((size_t)n, sizeof(uct_node));
                node->n_children = n;
                for (int k = 0; k < n; k++) node->children[k].action = legal[k];
            }
            double L = log((double)node->explore_count), best = -INFINITY;
            int bi = 0;
            for (int k = 0; k < node->n_children; k++) {
                const uct_node *c = &node->children[k];
                double v = c->solved ? (to_move == 0 ? c->outcome0 : -c->outcome0)
                           : c->explore_count == 0
                               ? INFINITY
                               : c->total_reward / (double)c->explore_count + uct_c * sqrt(L / (double)c->explore_count);
                if (v > best) {
                    best = v;
                    bi = k;
                }
            }
            node = &node->children[bi];
            orc_apply_action(&s, node->action);
            path[++depth] = node;
        }
        int solved = 0;
        if (s.terminal) {
            path[depth]->solved = 1;
            path[depth]->outcome0 = s.ret0;
            solved = 1;
        }
        while (!s.terminal) orc_apply_action(&s, random_legal_action(&s, r)); /* RandomRolloutEvaluator(1) */
        for (int d = depth; d >= 0; d--) {
            /* node at depth d >= 1 was entered by player (root_player + d - 1) & 1; the root carries the player to move */
            int mover = d == 0 ? root_player : (root_player + d + 1) & 1;
            uct_node *nd = path[d];
            nd->explore_count += 1;
            nd->total_reward += mover == 0 ? s.ret0 : -s.ret0;
            if (solved && nd->children) {
                const int player = (root_player + d) & 1; /* the player to move at nd */
                const uct_node *bst = NULL;
                int all_solved = 1;
                for (int k = 0; k < nd->n_children; k++) {
                    const uct_node *c = &nd->children[k];
                    if (!c->solved) all_solved = 0;
                    else {
                        double vc = player == 0 ? c->outcome0 : -c->outcome0;
                        if (!bst || vc > (player == 0 ? bst->outcome0 : -bst->outcome0)) bst = c;
                    }
                }
                if (bst && (all_solved || (player == 0 ? bst->outcome0 : -bst->outcome0) == 1.0)) {
                    nd->solved = 1;
                    nd->outcome0 = bst->outcome0;
                } else solved = 0;
            }
        }
        if (root.solved) break;
    }
    int bi = -1;
    double bo = 0;
    for (int k = 0; k < root.n_children; k++) {
        const uct_node *c = &root.children[k];
        double o = c->solved ? (root_player == 0 ? c->outcome0 : -c->outcome0) : 0.0;
        if (bi < 0 || o > bo || (o == bo && (c->explore_count > root.children[bi].explore_count ||
                                              (c->explore_count == root.children[bi].explore_count &&
                                               c->total_reward > root.children[bi].total_reward)))) {
            bi = k;
            bo = o;
        }
    }
    int action = bi >= 0 ? root.children[bi].action : -1;
    uct_free(&root);
    return action;
}

/* NeuralNetBot.step (alphazerobot.py:105-120): argmax (first maximum) of the masked, renormalised network priors */
static int net_bot_step(orc_policy_fn fn, void *user, const orc_state *s, int A, double *pol, double *board) {
    double v;
    orc_state_to_board(s, board);
    fn(user, s, board, pol, &v);
    int32_t legal[ORC_MAX_CELLS * 3];
    int nl = orc_legal_actions(s, legal);
    orc_remove_illegal_actions(pol, A, legal, nl);
    int action = 0;
    for (int a = 1; a < A; a++)
        if (pol[a] > pol[action]) action = a;
    return action;
}

#define ORC_ARENA_ZERO 1
#define ORC_ARENA_NET 2
#define ORC_OPPONENT_RANDOM 1
#define ORC_OPPONENT_UCT 2
typedef struct {
    int32_t game, rows, cols;
    int32_t n_playouts, keep_search_tree, agent, opponent, opponent_sims;
    double c_puct, temperature, opponent_uct_c;
    uint64_t seed;
    int32_t game_id; /* the agent plays side game_id & 1; the Philox stream is keyed by it */
    int32_t sample_plies; /* > 0: AlphaZeroBot(use_probabilistic_actions=True, num_probabilistic_actions=sample_plies)
                             (alphazerobot.py:34-36,81-86); 0: the greedy bot of the test_* pairings */
} orc_arena_cfg;

/* game_utils.play_game (game_utils.py:16-35) between the agent (AlphaZeroBot outside self-play, alphazerobot.py:42-93,
 * or NeuralNetBot, alphazerobot.py:96-120) and the opponent bot.  actions: the moves played; returns their count,
 * *ret0 = returns()[0]. */
int orc_play_arena_game(const orc_arena_cfg *cfg, orc_policy_fn fn, void *user, int32_t *actions, int max_actions, double *ret0) {
    orc_state s;
    orc_state_init(&s, cfg->game, cfg->rows, cfg->cols);
    int A = orc_num_actions(cfg->game, cfg->rows, cfg->cols);
    orc_bot *bot = cfg->agent == ORC_ARENA_ZERO
                       ? orc_bot_new(A, 0, cfg->keep_search_tree, cfg->temperature, cfg->c_puct, cfg->n_playouts, 0, 0.25, 1, fn, user)
                       : NULL;
    if (bot && cfg->sample_plies > 0) {
        bot->use_probabilistic_actions = 1;
        bot->num_probabilistic_actions = cfg->sample_plies;
    }
    double *pol = (double *)malloc(sizeof(double) * (size_t)A), *board = (double *)malloc(sizeof(double) * 4 * ORC_MAX_CELLS);
    int n = 0;
    while (!s.terminal) {
        int action;
        if ((s.nhist & 1) == (cfg->game_id & 1)) { /* the agent's turn */
            if (cfg->agent == ORC_ARENA_ZERO) {
                orc_philox ru; /* the uniform behind np.random.choice: the engine's move stream (purpose 1) */
                philox_init(&ru, cfg->seed, (uint32_t)cfg->game_id, (uint32_t)s.nhist, 1u, 0u);
                action = orc_bot_step(bot, &s, NULL, philox_u01(&ru), pol);
            } else {
                action = net_bot_step(fn, user, &s, A, pol, board);
            }
        } else {
            orc_philox r;
            philox_init(&r, cfg->seed, (uint32_t)cfg->game_id, (uint32_t)s.nhist, 2u, 0u);
            action = cfg->opponent == ORC_OPPONENT_RANDOM ? random_legal_action(&s, &r)
                                                           : uct_bot_action(&s, cfg->opponent_sims, cfg->opponent_uct_c, &r);
        }
        if (n >= max_actions || action < 0) {
            n = -1;
            break;
        }
        actions[n++] = action;
        orc_apply_action(&s, action);
    }
    *ret0 = s.ret0;
    if (bot) orc_bot_free(bot);
    free(pol);
    free(board);
    return n;
}

/* the opponent bots alone, for unit checks: the move they choose in `s` for (seed, game id) */
int orc_opponent_action(const orc_state *s, int opponent, int n_sims, double uct_c, uint64_t seed, int32_t game_id) {
    orc_philox r;
    philox_init(&r, seed, (uint32_t)game_id, (uint32_t)s->nhist, 2u, 0u);
    return opponent == ORC_OPPONENT_RANDOM ? random_legal_action(s, &r) : uct_bot_action(s, n_sims, uct_c, &r);
}


/* test_zero_vs_zero (game_utils.py:120-145) for ONE game: two AlphaZeroBots outside self-play, each with its own policy
 * function and settings; bot 1 plays side game_id & 1.  Root noise off (the device draws its noise from a fast-math gamma
 * sampler that a CPU cannot reproduce bit for bit; with noise the pairing is compared statistically). */
typedef struct {
    int32_t game, rows, cols, game_id;
    int32_t n_playouts1, n_playouts2, keep_search_tree, reserved;
    double c_puct1, c_puct2, temperature;
    int32_t agent1, agent2; /* ORC_ARENA_ZERO (AlphaZeroBot) or ORC_ARENA_NET (NeuralNetBot) */
} orc_duel_cfg;
int orc_play_duel_game(const orc_duel_cfg *cfg, orc_policy_fn fn1, orc_policy_fn fn2, void *user, int32_t *actions, int max_actions,
                       double *ret0) {
    orc_state s;
    orc_state_init(&s, cfg->game, cfg->rows, cfg->cols);
    int A = orc_num_actions(cfg->game, cfg->rows, cfg->cols);
    orc_bot *b1 = orc_bot_new(A, 0, cfg->keep_search_tree, cfg->temperature, cfg->c_puct1, cfg->n_playouts1, 0, 0.25, 1, fn1, user);
    orc_bot *b2 = orc_bot_new(A, 0, cfg->keep_search_tree, cfg->temperature, cfg->c_puct2, cfg->n_playouts2, 0, 0.25, 1, fn2, user);
    double *pol = (double *)malloc(sizeof(double) * (size_t)A), *board = (double *)malloc(sizeof(double) * 4 * ORC_MAX_CELLS);
    int n = 0;
    while (!s.terminal) {
        const int first = (s.nhist & 1) == (cfg->game_id & 1);
        orc_bot *b = first ? b1 : b2;
        int action = (first ? cfg->agent1 : cfg->agent2) == ORC_ARENA_NET ? net_bot_step(first ? fn1 : fn2, user, &s, A, pol, board)
                                                                        : orc_bot_step(b, &s, NULL, 0.0, pol);
        if (n >= max_actions) {
            n = -1;
            break;
        }
        actions[n++] = action;
        orc_apply_action(&s, action);
    }
    *ret0 = s.ret0;
    orc_bot_free(b1);
    orc_bot_free(b2);
    free(pol);
    free(board);
    return n;
}

int64_t orc_nodes_alive(void) { return g_nodes_alive; }
int64_t orc_nodes_total(void) { return g_nodes_total; }
