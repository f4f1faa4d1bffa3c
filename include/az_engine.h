/*
 * az_engine.h — C ABI of the MI355X-native AlphaZero self-play engine.
 *
 * This is the drop-in boundary for ONE path of danielwillemsen/alphazero-openspiel:
 * PUCT tree search (mcts.py) + self-play rollout loop (game_utils.py:148-206) +
 * agent step (alphazerobot.py:42-93) + game dynamics / state encoding
 * (pyspiel calls, network.py:9-18), run for G concurrent games on one GPU.
 * Leaf evaluation (network.py:48-64) is the caller's: the engine hands out a
 * device batch of observations and takes back device priors/values, so the
 * reference's multiprocess pipe protocol (examplegenerator.py:39-77) has no
 * equivalent here — its two messages are the two pointer arguments of
 * az_engine_advance().
 *
 * Conventions: every entry returns 0 on success, a negative AZ_E_* otherwise;
 * no C++ exception crosses this boundary; az_last_error() gives the text.
 * "dev" pointers are HIP device pointers owned by the caller; `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  All device work is
 * asynchronous on that stream unless a function says it synchronises.
 * The engine is not re-entrant: one host thread per engine, one engine (or
 * more) per GPU, one process per GPU.
 */
#ifndef AZ_ENGINE_H
#define AZ_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_GAME_CONNECT_FOUR 0 /* "connect_four"                      (train.py:24) */
#define AZ_GAME_BREAKTHROUGH 1 /* "breakthrough(rows=R,columns=C)"    (train.py:24, tournament.py) */

/* value-target modes of play_game_self (game_utils.py:166-194) */
#define AZ_BACKUP_ON_POLICY 0
#define AZ_BACKUP_SOFT_Z 1
#define AZ_BACKUP_A0C 2
#define AZ_BACKUP_OFF_POLICY 3

#define AZ_RNG_PHILOX 0   /* on-device Philox4x32-10 streams keyed by (seed, game id, ply) */
#define AZ_RNG_INJECTED 1 /* Dirichlet vectors and choice-uniforms supplied by the caller (parity mode) */

#define AZ_OK 0
#define AZ_E_INVALID (-1)   /* bad argument / config */
#define AZ_E_HIP (-2)       /* a HIP runtime call failed */
#define AZ_E_STATE (-3)     /* call order violated (e.g. advance before reset) */
#define AZ_E_DEVICE (-4)    /* a device-side fault flag is set (see az_progress.error_flags) */
#define AZ_E_NOMEM (-5)

/* device-side fault bits, OR-ed into az_progress.error_flags */
#define AZ_FAULT_POOL_EXHAUSTED 1u /* a slot's node pool is full even after compaction */
#define AZ_FAULT_PLY_OVERFLOW 2u   /* a game exceeded max_plies */
#define AZ_FAULT_NO_VISITS 4u      /* root has no visited child at move time (S too small) */
#define AZ_FAULT_BAD_PRIOR 8u      /* NaN prior/value fed to advance */
#define AZ_FAULT_ILLEGAL_ACTION 16u /* az_engine_update_root was given an action that is illegal in the slot's root state */
#define AZ_FAULT_VISIT_RANGE 32u    /* AZ_SELECT_UCT: a node's visit count left the log(N) table (N > n_playouts * (max_plies + 2)) */

#define AZ_ACTION_NONE (-1)         /* az_engine_update_root: leave the slot alone */
#define AZ_ACTION_SEARCH_AGAIN (-2) /* az_engine_update_root: search the same root again (MCTS.search called twice, mcts.py:164-180) */

/*
 * Search/agent configuration.  Field ↔ reference keyword (all reach the
 * reference's search as **kwargs: SURVEY.md §5 "Config / flag system"):
 *   n_playouts, c_puct, use_dirichlet, dirichlet_ratio   mcts.py:96-101
 *   temperature                                          alphazerobot.py:38
 *   keep_search_tree                                     alphazerobot.py:26,54-68
 *   backup                                               game_utils.py:155
 * dirichlet_alpha is the literal 0.3 of mcts.py:187; the noise weight is the
 * literal 0.25 of mcts.py:189 (NOT dirichlet_ratio) — reproduced as such.
 */
typedef struct az_config {
    int32_t struct_size; /* = sizeof(az_config); ABI guard */
    int32_t game;        /* AZ_GAME_* */
    int32_t rows, cols;  /* breakthrough board (connect_four: 6,7 enforced) */
    int32_t n_slots;     /* G: concurrent games resident on the device */
    int32_t n_playouts;  /* S */
    int32_t use_dirichlet;
    int32_t keep_search_tree;
    int32_t backup;            /* AZ_BACKUP_* */
    int32_t rng_mode;          /* AZ_RNG_* */
    int32_t max_sims_per_tick; /* bound on NN-free playouts (terminal hits) one slot chains per advance; 0 = default (10) */
    int32_t device;            /* HIP device ordinal */
    int32_t manual_moves;      /* 1: stop after each search (MCTS.search semantics); the caller reads the root and
                                  moves with az_engine_update_root — AlphaZeroBot.step outside self-play */
    int32_t chain_window_us;   /* a slot starts another NN-free playout only this early in a launch; 0 = default (10), <0 = off */
    int64_t nodes_per_slot;    /* node-pool capacity per slot; 0 = default from game and S */
    int64_t max_games;         /* capacity of the example store (games per reset) */
    double c_puct;
    double dirichlet_ratio;
    double dirichlet_alpha;
    double temperature;
    uint64_t seed;
    /* Evaluation arena (game_utils.py:16-145, examplegenerator.py:177-195, train.py:238-270): every game is played
     * between an AGENT driven by the network and an OPPONENT bot; game id i gives the agent the side i & 1, so the pair
     * (2k, 2k+1) is one `test_*_vs_*` call of the reference (agent first, then agent second).  0 / 0 = self-play. */
    int32_t arena_agent;    /* AZ_ARENA_* */
    int32_t arena_opponent; /* AZ_OPPONENT_* */
    int32_t opponent_sims;  /* AZ_OPPONENT_UCT: max_search_nodes of mcts.MCTSBot (game_utils.py:74-75) */
    int32_t arena_flip;     /* 1: the agent takes the OTHER side, (i & 1) ^ 1 - the second network of a two-engine pairing */
    double opponent_uct_c;  /* AZ_OPPONENT_UCT: uct_c (1 in the reference's calls) */
    /* MCTS(use_puct=...) (mcts.py:101,120): the rule of the trees that update_root STARTS when it finds a leaf root
     * (mcts.py:199-200).  The reference keeps the rule per Node, children inherit their parent's (mcts.py:64), and
     * MCTS.__init__ always builds a PUCT root (mcts.py:122): so a tree is PUCT unless it grew from a root created by
     * update_root on a leaf - reproduced here as one rule bit per slot tree.  0 (AZ_SELECT_PUCT) = use_puct=True. */
    int32_t select_rule; /* AZ_SELECT_* */
    /* AlphaZeroBot(use_probabilistic_actions=True) outside self-play (alphazerobot.py:34,83-84; tournament.py:35-36):
     * the AZ_ARENA_ZERO agent samples its move from the tempered visit distribution instead of taking the argmax. */
    int32_t arena_probabilistic;
    /* num_probabilistic_actions (alphazerobot.py:36,81-85): a sampling agent (self-play, or arena_probabilistic) samples
     * only while fewer than this many moves have been played, then plays the argmax.  0 = the reference's default, 1000; < 0 = never sample. */
    int32_t num_probabilistic_actions;
    /* node pools shared as compaction targets (a slot owns one pool of nodes_per_slot nodes; re-rooting copies the kept
     * subtree into a spare pool when the pool cannot hold another search): 0 = default (n_slots / 16, at least 16) */
    int32_t spare_pools;
} az_config;

#define AZ_SELECT_PUCT 0 /* Q + c_puct * P * sqrt(N_parent) / (N + 1)                         mcts.py:78 */
#define AZ_SELECT_UCT 1  /* inf if N == 0 else Q + c_puct * P * sqrt(log(N_parent) / N)       mcts.py:80 */

#define AZ_ARENA_SELF_PLAY 0
#define AZ_ARENA_ZERO 1 /* AlphaZeroBot outside self-play: search, then the most visited move (alphazerobot.py:86-91), tree kept
                           across both players' moves (alphazerobot.py:60-64) */
#define AZ_ARENA_NET 2  /* NeuralNetBot (alphazerobot.py:96-120): one evaluation, argmax of the masked, renormalised priors;
                           configure n_playouts = 1, use_dirichlet = 0, keep_search_tree = 0 */
#define AZ_OPPONENT_NONE 0
#define AZ_OPPONENT_RANDOM 1 /* pyspiel.make_uniform_random_bot: a uniformly random legal action */
#define AZ_OPPONENT_UCT 2    /* open_spiel.python.algorithms.mcts.MCTSBot(game, player, uct_c, max_search_nodes,
                                RandomRolloutEvaluator(1)) - third party, absent from the reference tree, version unpinned:
                                restated from its published algorithm (DESIGN_HISTORY.md section 6) */
#define AZ_OPPONENT_EXTERNAL 3 /* the opponent's moves come from ANOTHER engine (az_engine_exchange_moves): two AlphaZero agents with
                                  their own networks and settings - test_zero_vs_zero (game_utils.py:120-145) */

typedef struct az_sizes {
    int32_t num_actions;  /* A   = game.num_distinct_actions() */
    int32_t obs_planes;   /* C+1 = state_shape[0] + 1 (network.py:15) */
    int32_t rows, cols;   /* H, W */
    int32_t max_children; /* upper bound on legal actions of any state */
    int32_t max_plies;    /* upper bound on game length */
    int32_t n_slots;
    int32_t spare_pools;  /* pools beyond the slots' own: (n_slots + spare_pools) * nodes_per_slot nodes are allocated */
    int64_t nodes_per_slot;
    int64_t max_games;
    int64_t device_bytes; /* HBM held by the engine */
} az_sizes;

typedef struct az_progress {
    int64_t games_started, games_done;
    int64_t moves;         /* AlphaZeroBot.step calls completed */
    int64_t sims;          /* MCTS.playout calls completed */
    int64_t evals;         /* policy_fn requests issued (leaf + root-Dirichlet) */
    int64_t terminal_hits; /* playouts that ended on a terminal state (no evaluation) */
    int64_t sum_depth;     /* Σ select depth over playouts */
    int64_t sum_children;  /* Σ children scanned by select */
    int64_t nodes_allocated;
    int64_t compactions;
    int64_t slots_waiting; /* slots with an evaluation request outstanding after the last advance */
    int64_t slots_idle;
    int64_t slots_search_done; /* manual_moves: slots whose S playouts are complete */
    uint32_t error_flags; /* AZ_FAULT_* */
    uint32_t reserved;
} az_progress;

/* Host view of finished games, valid until the next reset/export/destroy.
 * Replaces the pickled list-of-games a pool returns (examplegenerator.py:151-152,172-173);
 * the Python façade turns each ply into the reference's [key, board, pi, z] record
 * (game_utils.py:169). */
typedef struct az_example_view {
    int64_t n_games;            /* finished games, ids 0..n_games-1 */
    int32_t max_plies;          /* row stride of the per-ply arrays */
    int32_t max_children;       /* row stride of counts/actions */
    const int32_t *game_len;    /* [n_games] plies */
    const float *game_ret0;     /* [n_games] returns()[0] */
    const uint64_t *states;     /* [n_games][max_plies][2] bitboards (layout: az_games.h) */
    const uint16_t *move;       /* [n_games][max_plies] action played */
    const uint8_t *n_children;  /* [n_games][max_plies] number of root children (= legal actions) */
    const uint16_t *child_action; /* [n_games][max_plies][max_children] ascending legal actions */
    const uint32_t *child_visits; /* [n_games][max_plies][max_children] root child N before the move */
    const double *value;        /* [n_games][max_plies] value target (soft-Z / A0C / off-policy; on-policy: filled from ret0) */
} az_example_view;

typedef struct az_engine az_engine;

/* lifecycle ------------------------------------------------------------- */
int az_engine_create(const az_config *cfg, az_engine **out);
int az_engine_destroy(az_engine *e);
const char *az_last_error(const az_engine *e); /* e may be NULL: error of the last failed create */
int az_engine_sizes(const az_engine *e, az_sizes *out);

/* Start a generation of n_games self-play games (ExampleGenerator.generate_examples(n_games),
 * examplegenerator.py:164-175).  Game ids are 0..n_games-1; id i uses RNG stream (seed, i).
 * Slots pick up ids in order and refill from a device-side counter as games finish. */
int az_engine_reset(az_engine *e, uint64_t seed, int64_t n_games, void *stream);

/* Parity mode (rng_mode = AZ_RNG_INJECTED): host arrays, copied.
 * etas [n_games][max_plies][max_children]: the np.random.dirichlet draw of each move (mcts.py:187)
 * us   [n_games][max_plies]: the uniform behind np.random.choice of each move (alphazerobot.py:84) */
int az_engine_set_injected_rng(az_engine *e, const double *etas, const double *us, int64_t n_games);

/* Start every slot from a given position instead of the initial one (test hook for mid-game search
 * traces).  actions: host [n] action prefix applied to the initial state. Call after reset. */
int az_engine_set_start_prefix(az_engine *e, const int32_t *actions, int32_t n);

/*
 * One tick = MCTS.playout's select + expand + backup (mcts.py:126-153) for all slots, fused with the
 * agent's move step when a slot has finished its S playouts (alphazerobot.py:71-93,
 * game_utils.py:156-197, mcts.py:155-162,192-203) and the root Dirichlet expansion (mcts.py:182-190):
 *   1. consume priors[g]/values[g] for the request slot g issued on the PREVIOUS tick
 *      (expand + backup, or root expansion); ignored for slots without a request;
 *   2. run playouts/moves until the slot needs the network again, and write that state's
 *      observation (state_to_board, network.py:9-18, float32) to obs_out[g].
 * priors: dev float32 [G][A] (softmax output), values: dev float32 [G], obs_out: dev float32 [G][C+1][H][W].
 * priors/values may be NULL on the first tick after reset.
 */
int az_engine_advance(az_engine *e, const float *priors, const float *values, float *obs_out, void *stream);

/* The same tick for the slots [first_slot, first_slot + n_slots) only.  priors / values / obs_out are still the
 * WHOLE-engine arrays ([G][A], [G], [G][C+1][H][W]); the launch reads and writes the rows of its slots.  Slot groups
 * are independent (they share only the game-id counter, by device atomics), so disjoint groups may be advanced
 * concurrently on DIFFERENT streams: while one group's PV-net forward runs, another group's tree search runs beside it
 * (BASELINE.json configs[4]: "overlapped PV-eval / tree-search HIP streams"; the reference's analogue is its pool of
 * worker processes running beside the inference server, examplegenerator.py:106-138).  Which group plays which game
 * does not change a game: random streams are keyed by game id.  az_engine_progress / poll / export synchronise only the
 * stream they are given: synchronise the other group streams first.  priors/values must not be NULL here (give
 * initialised buffers on the first tick: slots without an outstanding request ignore them). */
int az_engine_advance_slots(az_engine *e, int32_t first_slot, int32_t n_slots, const float *priors, const float *values,
                            float *obs_out, void *stream);

/* Arena engines only: compute the opponent bot's move for every slot whose opponent is to move (one thread per slot:
 * a uniformly random legal action, or a UCT search with random rollouts).  The next az_engine_advance applies the moves.
 * One arena tick = az_engine_advance, az_engine_opponent_moves, PV-net forward.  Asynchronous on `stream`. */
int az_engine_opponent_moves(az_engine *e, void *stream);

/* The tail of a generation.  Once every game has been handed to a slot, finished slots stay idle and the batch thins out while a
 * tick still spans all n_slots rows of the request buffers (a generation lasts as long as its longest game).
 * az_engine_compact_rows (synchronises `stream`) lists the slots that still play, densely and in slot order, and returns their
 * number; from then on the engine is ticked with az_engine_advance_rows over n_rows >= that number rows: list entry i writes its
 * request to row i of obs_out and finds the answer to its previous request in the row it was written to, so the network only has to
 * evaluate the first n_rows rows (`az_net_forward(..., n_rows, ...)`).  May be called again as the list thins further; refused
 * while games are still being handed out, and on arena / manual_moves engines.  az_engine_reset returns to one row per slot.
 * Replaces nothing in the reference (its worker processes simply exit, examplegenerator.py:134-138). */
int az_engine_compact_rows(az_engine *e, int32_t *n_live_out, void *stream);
int az_engine_advance_rows(az_engine *e, int32_t n_rows, const float *priors, const float *values, float *obs_out, void *stream);

/* Two arena engines facing each other (both AZ_OPPONENT_EXTERNAL, same game, same n_slots, arena_flip 0 and 1, games = slots:
 * no refill): for every slot, hand a's agent move to b and b's to a as soon as it has been played.  One tick of such a
 * pairing = advance(a), advance(b), az_engine_exchange_moves(a, b), forward(net of a), forward(net of b). */
int az_engine_exchange_moves(az_engine *a, az_engine *b, void *stream);

/* MCTS.update_root(action) (mcts.py:192-203) for every slot, manual_moves engines only: applies
 * actions[g] (host array [G]; AZ_ACTION_NONE = leave the slot alone) to the slot's root state, keeps the chosen
 * child's subtree (or starts a fresh tree when keep_subtree == 0 or the root is a leaf) and arms the
 * next search.  A slot whose game ends goes idle.  An action that is illegal in the root state raises
 * AZ_FAULT_ILLEGAL_ACTION and idles the slot (the state is left untouched).  AZ_ACTION_SEARCH_AGAIN re-arms a finished
 * search on the SAME root: the tree is kept, another n_playouts run, the root is re-expanded with a fresh Dirichlet
 * draw - what calling MCTS.search(state) twice does in the reference (mcts.py:164-190). */
int az_engine_update_root(az_engine *e, const int32_t *actions, int32_t keep_subtree, void *stream);

/* Counters; synchronises `stream`. */
int az_engine_progress(az_engine *e, az_progress *out, void *stream);

/* Cheap completion poll for the tick loop: number of finished games and the device fault flags (two words, one
 * small copy); synchronises `stream`.  Returns AZ_E_DEVICE if a fault flag is set. */
int az_engine_poll(az_engine *e, int64_t *games_done, uint32_t *error_flags, void *stream);

/* Copy finished games to host memory owned by the engine; synchronises `stream`. */
int az_engine_export(az_engine *e, az_example_view *out, void *stream);

/* The same records, device to device: the finished games of the generation packed into ONE caller-owned device buffer
 * (asynchronous on `stream`), ready for the generation-end collective (RCCL all-gather over xGMI) and for
 * az_replay_append_device on the receiving side - the replacement of the pickled `pool.map_async(...).get()` gather of
 * examplegenerator.py:151-152 without a host round trip.  Layout, n = games of the generation (the n_games given to
 * az_engine_reset), mp = max_plies, mc = max_children, every array 16-byte aligned, in this order:
 *   game_len i32[n] | game_ret0 f32[n] | states u64[n][mp][2] | move u16[n][mp] | n_children u8[n][mp] |
 *   child_action u16[n][mp][mc] | child_visits u32[n][mp][mc] | value f64[n][mp]
 * (the arrays of az_example_view; on-policy value targets are filled in on the device). */
int64_t az_engine_export_device_bytes(const az_engine *e);
int az_engine_export_device(az_engine *e, void *dev_buf, int64_t bytes, void *stream);

/* debug / parity read-back of one slot's root (mcts.root.{N,Q,children[a].{N,Q,P}}, read by
 * game_utils.py:30-31,174,178,183-193).  Arrays sized max_children.  Returns n_children or <0.
 * Synchronises the device. */
int az_engine_read_root(az_engine *e, int32_t slot, int64_t *root_n, double *root_q, int32_t *actions,
                        int64_t *child_n, double *child_q, double *child_p);

/* The whole search tree of one slot in breadth-first order (node 0 = root; children of a node are
 * consecutive and in ascending-action order): parent index, action leading to the node, N, Q, P.
 * What deep-copying `mcts.root` gives the reference's statistics code (game_utils.py:30-31,183-193).
 * Arrays sized max_nodes (any may be NULL).  Returns the node count, or <0; if the tree has more than max_nodes
 * nodes the first max_nodes are written and the full count is returned.  Synchronises the device. */
int64_t az_engine_read_tree(az_engine *e, int32_t slot, int64_t max_nodes, int32_t *parent, int32_t *action,
                            int64_t *n, double *q, double *p);

typedef struct az_slot_info {
    int32_t phase; /* 0 idle, 1 run, 2 move pending, 3 waiting root eval, 4 waiting leaf eval, 5 search done (manual_moves) */
    int32_t game_id, ply, sims_done;
    uint32_t root, alloc;
    uint64_t bb[2];      /* root state */
    uint64_t leaf_bb[2]; /* state of the outstanding request */
    int32_t leaf_ply, depth;
} az_slot_info;
int az_engine_read_slot(az_engine *e, int32_t slot, az_slot_info *out);

#ifdef __cplusplus
}
#endif
#endif /* AZ_ENGINE_H */
