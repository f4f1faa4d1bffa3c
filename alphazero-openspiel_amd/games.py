"""Host-side game objects with the (pre-2020) OpenSpiel `pyspiel` surface the reference is written
against — `load_game`, `Game.new_initial_state`, `State.{clone, current_player, is_terminal,
apply_action, legal_actions, player_return, returns, history, information_state,
information_state_as_normalized_vector}` (call sites: reference mcts.py:138-149,178,184;
alphazerobot.py:29,55,72; game_utils.py:150-153,156,169,197,201; network.py:15-17).

They use the SAME bitboard layouts as the device code (csrc/az_games.h), so a state can be handed to
the engine (as its action history) and engine records (bitboards) can be turned back into boards:

  connect_four : bit = col*7 + row (row 0 = bottom); bb[0] player-0 'x', bb[1] player-1 'o'
  breakthrough : bit = row*C + col; bb[0] black (player 0, rows 0..1, moves up), bb[1] white

This module is host logic only (no search, no evaluation).  The self-play hot path runs on the GPU.
"""
import numpy as np

TERMINAL_PLAYER = -4
_M64 = (1 << 64) - 1


def parse_game_name(name):
    """'connect_four' | 'breakthrough(rows=R,columns=C)' -> (game_id, rows, cols)   (train.py:24)"""
    name = name.strip()
    if name in ("connect_four", "connect_four()"):
        return 0, 6, 7
    if name.startswith("breakthrough"):
        rows = cols = 8
        if "(" in name:
            for kv in filter(None, (s.strip() for s in name[name.index("(") + 1:name.rindex(")")].split(","))):
                k, v = kv.split("=")
                if k.strip() == "rows":
                    rows = int(v)
                elif k.strip() == "columns":
                    cols = int(v)
        return 1, rows, cols
    raise ValueError("unsupported game %r (connect_four and breakthrough are built)" % name)


class Game:
    def __init__(self, name):
        self.game_id, self.rows, self.cols = parse_game_name(name)
        self.name = "connect_four" if self.game_id == 0 else "breakthrough(rows=%d,columns=%d)" % (self.rows, self.cols)

    def num_players(self):
        return 2

    def num_distinct_actions(self):
        return 7 if self.game_id == 0 else self.rows * self.cols * 12

    def information_state_normalized_vector_shape(self):
        return [3, self.rows, self.cols]

    observation_tensor_shape = information_state_normalized_vector_shape

    def max_game_length(self):
        return 42 if self.game_id == 0 else 2 * self.cols * (2 * self.rows - 5) + 1

    def max_children(self):
        return 7 if self.game_id == 0 else min(64, 6 * self.cols)

    def new_initial_state(self):
        return State(self)

    def __str__(self):
        return self.name


def load_game(name):
    return Game(name)


def _c4_has_four(b):
    for s in (7, 6, 8, 1):
        m = b & (b >> s)
        if m & (m >> (2 * s)):
            return True
    return False


class State:
    def __init__(self, game):
        self._game = game
        self._hist = []
        self._ret0 = None
        if game.game_id == 0:
            self.bb = [0, 0]
        else:
            two = (1 << (2 * game.cols)) - 1
            self.bb = [two, two << ((game.rows - 2) * game.cols)]

    # -- protocol ----------------------------------------------------------------------------
    def clone(self):
        s = State.__new__(State)
        s._game, s._hist, s._ret0, s.bb = self._game, self._hist[:], self._ret0, self.bb[:]
        return s

    def get_game(self):
        return self._game

    def current_player(self):
        return TERMINAL_PLAYER if self._ret0 is not None else len(self._hist) & 1

    def is_terminal(self):
        return self._ret0 is not None

    def history(self):
        return self._hist[:]

    def returns(self):
        z = 0.0 if self._ret0 is None else self._ret0
        return [z, -z]

    def player_return(self, player):
        return self.returns()[player]

    def information_state(self, player=None):
        return ", ".join(str(a) for a in self._hist)

    information_state_string = information_state

    def legal_actions(self, player=None):
        if self._ret0 is not None:
            return []
        return legal_actions_from_bitboards(self._game, self.bb[0], self.bb[1], len(self._hist))

    def apply_action(self, action):
        g = self._game
        action = int(action)
        if self._ret0 is not None:
            raise ValueError("apply_action on a terminal state")
        if action not in self.legal_actions():
            raise ValueError("illegal action %d" % action)
        me = len(self._hist) & 1
        if g.game_id == 0:
            occ = self.bb[0] | self.bb[1]
            nb = (occ | (occ + (1 << (action * 7)))) ^ occ
            self.bb[me] |= nb
            self._hist.append(action)
            if _c4_has_four(self.bb[me]):
                self._ret0 = 1.0 if me == 0 else -1.0
            elif len(self._hist) == 42:
                self._ret0 = 0.0
        else:
            d, cell = (action >> 1) % 6, (action >> 1) // 6
            t = cell + (-g.cols if me else g.cols) + (d % 3) - 1
            self.bb[me] = (self.bb[me] ^ (1 << cell)) | (1 << t)
            self.bb[1 - me] &= ~(1 << t) & _M64
            self._hist.append(action)
            if t // g.cols == (0 if me else g.rows - 1) or self.bb[1 - me] == 0:
                self._ret0 = 1.0 if me == 0 else -1.0

    def information_state_as_normalized_vector(self, player=None):
        return observation_planes(self._game, np.array([self.bb], dtype=np.uint64))[0].reshape(-1).tolist()

    observation_tensor = information_state_as_normalized_vector

    def __str__(self):
        g = self._game
        planes = observation_planes(g, np.array([self.bb], dtype=np.uint64))[0]
        if g.game_id == 0:
            ch = ".ox"
            idx = planes.argmax(0)
            return "\n".join("".join(ch[v] for v in row) for row in idx[::-1])
        ch = "bw."
        return "\n".join("".join(ch[v] for v in row) for row in planes.argmax(0))


def legal_actions_from_bitboards(game, bb0, bb1, ply):
    """Ascending legal actions of the side to move (ply & 1)."""
    if game.game_id == 0:
        occ = bb0 | bb1
        return [c for c in range(7) if not (occ >> (c * 7 + 5)) & 1]
    R, C = game.rows, game.cols
    me = ply & 1
    own, opp = (bb1, bb0) if me else (bb0, bb1)
    out = []
    for cell in range(R * C):
        if not (own >> cell) & 1:
            continue
        r, c = divmod(cell, C)
        r2 = r - 1 if me else r + 1
        if not 0 <= r2 < R:
            continue
        for d in range(3):
            c2 = c + d - 1
            if not 0 <= c2 < C:
                continue
            t = r2 * C + c2
            if (own >> t) & 1:
                continue
            is_opp = (opp >> t) & 1
            if d == 1 and is_opp:
                continue
            out.append(((cell * 6 + (3 if me else 0) + d) << 1) | int(is_opp))
    return out


def _cell_bits(game, bbs):
    """uint64 [n,2] bitboards -> two uint8 [n, H*W] arrays (player 0's / player 1's pieces) in the tensor's row-major cell order."""
    bbs = np.asarray(bbs, dtype=np.uint64).reshape(-1, 2)
    R, C = game.rows, game.cols
    if game.game_id == 0:
        rows, cols = np.meshgrid(np.arange(6), np.arange(7), indexing="ij")
        shift = (cols * 7 + rows).astype(np.uint64).reshape(-1)
    else:
        shift = np.arange(R * C, dtype=np.uint64)
    one = np.uint64(1)
    return (((bbs[:, 0:1] >> shift[None, :]) & one).astype(np.uint8), ((bbs[:, 1:2] >> shift[None, :]) & one).astype(np.uint8))


def _fill_planes(game, out, p0, p1):
    """out float64 [n, >=3, H*W]: the observation planes (the old OpenSpiel layout the shipped checkpoints pin:
    connect_four [empty, player-1, player-0]; breakthrough [black, white, empty])."""
    empty = 1 - p0 - p1
    planes = (empty, p1, p0) if game.game_id == 0 else (p0, p1, empty)
    for k, pl in enumerate(planes):
        out[:, k, :] = pl  # one cast-and-store pass per plane, no float temporaries


def observation_planes(game, bbs):
    """bbs: uint64 [n,2] -> float64 [n,3,H,W] observation planes."""
    p0, p1 = _cell_bits(game, bbs)
    out = np.empty((p0.shape[0], 3, game.rows * game.cols), dtype=np.float64)
    _fill_planes(game, out, p0, p1)
    return out.reshape(-1, 3, game.rows, game.cols)


def boards_from_bitboards(game, bbs, plies):
    """state_to_board (reference network.py:9-18) for many recorded states at once:
    float64 [n,4,H,W], last plane = player to move (ply & 1)."""
    p0, p1 = _cell_bits(game, bbs)
    out = np.empty((p0.shape[0], 4, game.rows * game.cols), dtype=np.float64)
    _fill_planes(game, out, p0, p1)
    out[:, 3, :] = (np.asarray(plies).reshape(-1) & 1)[:, None]
    return out.reshape(-1, 4, game.rows, game.cols)


def state_from_history(game, history):
    s = game.new_initial_state()
    for a in history:
        s.apply_action(a)
    return s
