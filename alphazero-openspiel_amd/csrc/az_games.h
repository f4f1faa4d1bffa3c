// az_games.h — bitboard game dynamics for connect_four and breakthrough(R×C), host+device.
//
// Replaces, for the engine, the OpenSpiel (pyspiel) state API the reference's hot path calls
// (mcts.py:138-149,178,184; alphazerobot.py:55,72; game_utils.py:150-153,156,169,197,201;
// network.py:15-17): apply_action / legal_actions / is_terminal / returns / current_player and
// the observation tensor.  OpenSpiel is absent from the reference tree (third-party, unpinned);
// rules follow the public game definitions, plane order and action codec follow the pins the
// shipped checkpoints give.
//
// Bit layouts
//   connect_four : bit = col*7 + row, row 0 = bottom, bit 6 of every column is a zero sentinel.
//                  bb[0] = player-0 ('x') stones, bb[1] = player-1 ('o') stones.
//   breakthrough : bit = row*C + col, bb[0] = black (player 0, home rows 0..1, moves to higher
//                  rows), bb[1] = white (player 1).  R*C <= 64.
// Player to move = ply & 1 in both games (strict alternation, no passes).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AZ_HD __host__ __device__ __forceinline__
#else
#define AZ_HD inline
#endif

#define AZG_CONNECT_FOUR 0
#define AZG_BREAKTHROUGH 1

struct AzState {
    uint64_t bb0, bb1;
    int ply;
};

struct AzGeom { // wave-uniform constants of the board
    int rows, cols, cells;
    uint64_t board_mask; // breakthrough: all cells
    uint64_t not_col0, not_col_last;
    uint64_t row_last, row_first; // breakthrough goal rows for black / white
};

AZ_HD int az_popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

static inline AzGeom az_make_geom(int game, int rows, int cols) {
    AzGeom g;
    g.rows = rows;
    g.cols = cols;
    g.cells = rows * cols;
    g.board_mask = g.cells >= 64 ? ~0ull : ((1ull << g.cells) - 1);
    g.not_col0 = g.not_col_last = g.row_last = g.row_first = 0;
    if (game == AZG_BREAKTHROUGH)
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) {
                uint64_t b = 1ull << (r * cols + c);
                if (c > 0) g.not_col0 |= b;
                if (c < cols - 1) g.not_col_last |= b;
                if (r == rows - 1) g.row_last |= b;
                if (r == 0) g.row_first |= b;
            }
    return g;
}

template <int GAME> AZ_HD void az_init_state(AzState &s, const AzGeom &g) {
    s.ply = 0;
    if (GAME == AZG_CONNECT_FOUR) {
        s.bb0 = s.bb1 = 0;
    } else {
        uint64_t two_rows = (g.cols * 2 >= 64) ? ~0ull : ((1ull << (2 * g.cols)) - 1);
        s.bb0 = two_rows;
        s.bb1 = two_rows << ((g.rows - 2) * g.cols);
    }
}

// ---------------------------------------------------------------- connect_four
AZ_HD bool az_c4_has_four(uint64_t b) {
    uint64_t m = b & (b >> 7);
    if (m & (m >> 14)) return true; // horizontal
    m = b & (b >> 6);
    if (m & (m >> 12)) return true; // diagonal
    m = b & (b >> 8);
    if (m & (m >> 16)) return true; // anti-diagonal
    m = b & (b >> 1);
    return (m & (m >> 2)) != 0; // vertical
}

// 7-bit mask of playable columns (ascending column = ascending action)
AZ_HD uint32_t az_c4_legal_mask(const AzState &s) {
    uint64_t occ = s.bb0 | s.bb1;
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 7; c++) m |= (uint32_t)(((occ >> (c * 7 + 5)) & 1ull) ^ 1ull) << c;
    return m;
}

// ---------------------------------------------------------------- breakthrough
// action = ((cell*6) + dir)*2 + capture; dir 0..2 black (+1 row; dc=-1,0,+1), 3..5 white (-1 row)
// Per-cell legality of the 3 forward moves of the side to move; bit d of the result = dir (d0+d) legal,
// bit (4+d) = that move is a capture.
AZ_HD uint32_t az_bt_cell_moves(const AzState &s, const AzGeom &g, int cell) {
    int me = s.ply & 1;
    uint64_t own = me ? s.bb1 : s.bb0, opp = me ? s.bb0 : s.bb1;
    uint64_t bit = 1ull << cell;
    if (!(own & bit)) return 0;
    uint32_t out = 0;
    int fwd = me ? -g.cols : g.cols;
    bool row_ok = me ? (cell >= g.cols) : (cell + g.cols < g.cells);
    if (!row_ok) return 0;
    // dc = -1
    if (g.not_col0 & bit) {
        uint64_t t = 1ull << (cell + fwd - 1);
        if (!(own & t)) out |= 1u | ((opp & t) ? 16u : 0u);
    }
    { // dc = 0: only onto an empty cell
        uint64_t t = 1ull << (cell + fwd);
        if (!((own | opp) & t)) out |= 2u;
    }
    if (g.not_col_last & bit) {
        uint64_t t = 1ull << (cell + fwd + 1);
        if (!(own & t)) out |= 4u | ((opp & t) ? 64u : 0u);
    }
    return out;
}

AZ_HD int az_bt_encode(int cell, int me, int d, int capture) { return ((cell * 6 + (me ? 3 : 0) + d) << 1) | capture; }

// ---------------------------------------------------------------- common
// Applies `action` for the side to move.  Returns 0 = game continues, 1 = terminal; *ret0 = returns()[0].
template <int GAME> AZ_HD int az_apply(AzState &s, const AzGeom &g, int action, float *ret0) {
    int me = s.ply & 1;
    if (GAME == AZG_CONNECT_FOUR) {
        uint64_t occ = s.bb0 | s.bb1;
        uint64_t nb = (occ | (occ + (1ull << (action * 7)))) ^ occ;
        uint64_t mine = (me ? s.bb1 : s.bb0) | nb;
        if (me) s.bb1 = mine; else s.bb0 = mine;
        s.ply++;
        if (az_c4_has_four(mine)) {
            *ret0 = me ? -1.f : 1.f;
            return 1;
        }
        if (s.ply == 42) {
            *ret0 = 0.f;
            return 1;
        }
        return 0;
    } else {
        int d = (action >> 1) % 6, cell = (action >> 1) / 6;
        int dc = (d % 3) - 1;
        int t = cell + (me ? -g.cols : g.cols) + dc;
        uint64_t from = 1ull << cell, to = 1ull << t;
        uint64_t own = me ? s.bb1 : s.bb0, opp = me ? s.bb0 : s.bb1;
        own = (own ^ from) | to;
        opp &= ~to;
        if (me) { s.bb1 = own; s.bb0 = opp; } else { s.bb0 = own; s.bb1 = opp; }
        s.ply++;
        if ((to & (me ? g.row_first : g.row_last)) || opp == 0) {
            *ret0 = me ? -1.f : 1.f;
            return 1;
        }
        return 0;
    }
}

// Scalar (one thread = one state) legal-action helpers, ascending action order = OpenSpiel's legal_actions() order.
// Used by the arena's opponent bots (random rollouts), where a thread owns a whole game.
template <int GAME> AZ_HD int az_count_legal(const AzState &s, const AzGeom &g) {
    if (GAME == AZG_CONNECT_FOUR) return az_popc64((uint64_t)az_c4_legal_mask(s));
    int me = s.ply & 1, n = 0;
    uint64_t own = me ? s.bb1 : s.bb0;
    while (own) {
        int cell = az_popc64((own & (0 - own)) - 1);
        own &= own - 1;
        n += az_popc64((uint64_t)(az_bt_cell_moves(s, g, cell) & 7u));
    }
    return n;
}
// the n-th (0-based) legal action in ascending order; n must be < az_count_legal
template <int GAME> AZ_HD int az_nth_legal(const AzState &s, const AzGeom &g, int n) {
    if (GAME == AZG_CONNECT_FOUR) {
        uint32_t m = az_c4_legal_mask(s);
        for (int c = 0; c < 7; c++)
            if ((m >> c) & 1u) {
                if (n == 0) return c;
                n--;
            }
        return -1;
    }
    int me = s.ply & 1;
    uint64_t own = me ? s.bb1 : s.bb0;
    while (own) {
        int cell = az_popc64((own & (0 - own)) - 1);
        own &= own - 1;
        uint32_t mv = az_bt_cell_moves(s, g, cell);
        for (int d = 0; d < 3; d++)
            if (mv & (1u << d)) {
                if (n == 0) return az_bt_encode(cell, me, d, (mv >> (4 + d)) & 1u);
                n--;
            }
    }
    return -1;
}

// Observation element idx of state_to_board's (C+1,H,W) tensor (network.py:9-18), C = 3.
//   connect_four planes: 0 empty, 1 player-1 stones, 2 player-0 stones, 3 current player
//   breakthrough planes: 0 black,  1 white,          2 empty,           3 current player
template <int GAME> AZ_HD float az_obs_elem(const AzState &s, const AzGeom &g, int idx) {
    // (no run-time division: this sits at the end of every playout, on the tick kernel's critical path)
    const int cells = GAME == AZG_CONNECT_FOUR ? 42 : g.cells;
    int plane = (idx >= cells) + (idx >= 2 * cells) + (idx >= 3 * cells), cell = idx - plane * cells;
    if (plane == 3) return (float)(s.ply & 1);
    uint64_t bit;
    if (GAME == AZG_CONNECT_FOUR) {
        int row = cell / 7, col = cell - row * 7;
        bit = 1ull << (col * 7 + row);
        uint64_t sel = plane == 0 ? ~(s.bb0 | s.bb1) : (plane == 1 ? s.bb1 : s.bb0);
        return (sel & bit) ? 1.f : 0.f;
    } else {
        bit = 1ull << cell;
        uint64_t sel = plane == 0 ? s.bb0 : (plane == 1 ? s.bb1 : ~(s.bb0 | s.bb1));
        return (sel & bit) ? 1.f : 0.f;
    }
}

static inline int az_num_actions(int game, int rows, int cols) { return game == AZG_CONNECT_FOUR ? 7 : rows * cols * 12; }
static inline int az_max_children(int game, int rows, int cols) {
    if (game == AZG_CONNECT_FOUR) return 7;
    int m = 6 * cols; // 2*cols pieces x 3 directions
    return m > 64 ? 64 : m;
}
static inline int az_max_plies(int game, int rows, int cols) {
    return game == AZG_CONNECT_FOUR ? 42 : 2 * cols * (2 * rows - 5) + 1;
}
