// az_engine_internal.h — definitions shared by az_engine.hip and az_replay.hip (NOT part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/az_engine.h"
#include "az_games.h"

#define NONE32 0xFFFFFFFFu

enum { PH_IDLE = 0, PH_RUN = 1, PH_MOVE = 2, PH_WAIT_ROOT = 3, PH_WAIT_LEAF = 4, PH_SEARCH_DONE = 5, PH_NEED_ROOT = 6,
       PH_OPPONENT = 7 /* arena: the opponent bot is to move */, PH_OPP_DONE = 8 /* ... and has chosen (opp_action) */ };
enum { ST_MOVES = 0, ST_SIMS, ST_EVALS, ST_TERM, ST_DEPTH, ST_CHILDREN, ST_NODES, ST_COMPACT, ST_N };

// One search-tree node (mcts.py:10-20 Node: N, Q, P, children).  32 bytes, so a block of sibling nodes is one
// contiguous run and a lane fetches its child with two 16-byte loads.
struct __attribute__((aligned(32))) AzNode {
    uint32_t N;    // visit count
    uint32_t C0;   // index of the first child in the slot's pool half (NONE32 = leaf)
    uint32_t META; // action leading here | n_children << 16
    uint32_t pad;
    double Q, P;
};

struct PwPlan { // numpy pairwise-sum recursion for a length-A vector, flattened (see np_sum_sparse)
    int n_blocks;
    int lo[16], len[16];
    int n_ops;
    int ops[32]; // >=0: push block i, -1: add top two
};

struct Params {
    // geometry / config
    AzGeom geom;
    int game, A, maxc, max_plies, obs_elems, pstride;
    int G, S, use_dirichlet, keep_tree, backup, rng_mode, max_sims_per_tick, manual_moves, chain_clocks;
    uint32_t cap, need_per_move;
    double c_puct, one_minus_ratio, alpha, inv_temp;
    uint64_t seed;
    long long n_games, max_games;
    AzState start;
    PwPlan pw;
    // node pools (array of structures: a child block is one contiguous run of 32-byte nodes): G + n_spare pools of `cap` nodes.
    // A slot OWNS one pool (which[g] & POOL_MASK); a compaction copies the kept subtree into a spare pool taken from
    // spare[] (entry = pool id, -1 = taken) and hands the old pool back through the same entry.
    AzNode *nodes;
    int n_spare;
    int *spare;
    // handed-over compaction: a slot that must compact while re-rooting takes a spare pool, becomes its owner (root 0) and
    // publishes a job once its own state is stored; the extra workgroups of the SAME launch copy the subtree, a workgroup per
    // job, and leave only when every slot wave of the launch has finished and the list is drained.  Nothing about a job outlives
    // its launch (no host-side parity or epoch: a captured graph replays any number of launches).
    int defer_compact;              // this launch carries the extra workgroups (set per launch by the host)
    int *cj_job, *cj_seen;          // [G] by ROW of the launch: slot + 1 whose copy the row's wave handed over (0: none); the epoch in
                                    // which the row's wave last finished
    int *cjob_count;                // [0] = the epoch of the running launch (advanced by the last extra workgroup out), [2] = extra
                                    // workgroups finished
    int *cj_from, *cj_entry;        // [G] the pool the subtree is copied out of; the spare[] entry that takes it back
    uint32_t *cj_root;              // [G] index of the new root in that pool
    // per slot
    int *phase, *gid, *ply, *sims, *which, *depth, *leaf_ply;
    uint32_t *root, *alloc, *leaf_node, *path;
    uint64_t *bb0, *bb1, *leaf_bb0, *leaf_bb1;
    unsigned long long *stats; // [G][ST_N]
    // dense rows (tail of a generation): row_slot[i] = the i-th slot that still plays, req_row[g] = the row of the request /
    // answer buffers that holds slot g's outstanding request; n_rows_live (device) = length of the list
    int *row_slot, *req_row, *n_rows_live;
    // global counters
    unsigned long long *next_game, *games_done;
    unsigned int *faults;
    // injected randomness
    const double *etas, *us;
    double *eta_buf; // [G][maxc] Dirichlet draw staged with the root request (Philox mode)
    // arena (evaluation games against a bot)
    int arena_agent, opp_kind, opp_sims, arena_flip;
    double opp_c;
    int *opp_action;           // [G] the opponent bot's chosen move
    uint32_t uct_cap;          // UCT opponent: nodes per slot
    uint32_t *uct_N, *uct_C0, *uct_META; // [G][uct_cap] explore_count, first child, action | n_children << 16
    double *uct_W;             // [G][uct_cap] total_reward, seen by the player who moved into the node
    const double *log_table;   // [log_n] log(n) computed on the host (glibc), so that host and device agree bit for bit
    uint32_t log_n;
    int arena_prob, n_prob_plies; // use_probabilistic_actions outside self-play; num_probabilistic_actions (alphazerobot.py:34-36)
    int select_rule;           // AZ_SELECT_*: rule of the trees update_root starts from a leaf root (mcts.py:199-200)
    // records
    int *rec_len;
    float *rec_ret0;
    uint64_t *rec_states;
    uint16_t *rec_move, *rec_child_action;
    uint8_t *rec_nchild;
    uint32_t *rec_child_visits;
    double *rec_value;
};


struct az_engine {
    az_config cfg;
    Params p;
    az_sizes sizes;
    std::string err;
    std::vector<void *> dev_allocs;
    bool reset_done = false;
    int64_t n_games = 0;
    double *d_etas = nullptr, *d_us = nullptr;
    int *d_actions = nullptr;
    int64_t ticks = 0;
    int64_t inj_games = 0;
    bool rows_mapped = false; // az_engine_compact_rows has been called since the last reset
    int rows_live = 0;
    bool may_compact = false; // a pool cannot hold a whole game: re-rooting may have to compact (launches carry extra workgroups)
    // host mirrors for export
    std::vector<int32_t> h_len;
    std::vector<float> h_ret0;
    std::vector<uint64_t> h_states;
    std::vector<uint16_t> h_move, h_child_action;
    std::vector<uint8_t> h_nchild;
    std::vector<uint32_t> h_child_visits;
    std::vector<double> h_value;
};

