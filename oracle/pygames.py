"""TEST INFRASTRUCTURE — pure-Python game objects used to DRIVE the real reference.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import anything under `oracle/`.  The product package never does.

These classes implement the (pre-2020) OpenSpiel state/game protocol that the
reference's hot path duck-types against (call sites: /root/reference/mcts.py:138-149,
178,184; alphazerobot.py:29,55,72; game_utils.py:150-153,156,169,197,201;
network.py:15-17).  OpenSpiel itself is a third-party dependency that is absent
from /root/reference and from this image (no pinned version exists: the
reference has no requirements file), so the RULES below are restated from the
public game definitions and are "parity unpinned" against OpenSpiel.  The
observation-plane order and the breakthrough action codec ARE pinned — by the
two checkpoints the reference ships; tests/test_checkpoint_pins.py
re-runs that experiment.

Deliberately written with plain 2-D cell arrays (no bitboards) so that it is an
independent implementation from both the C restatement (oracle/az_oracle.c) and
the HIP engine's bitboard device functions.
"""

TERMINAL_PLAYER = -4  # OpenSpiel's kTerminalPlayerId


class _Game:
    def num_players(self):
        return 2

    def new_initial_state(self):
        raise NotImplementedError


# --------------------------------------------------------------------------
# connect_four: 6 rows x 7 cols, row 0 = bottom row, action = column.
# cell values: 0 empty, 1 = player-1 piece ('o'), 2 = player-0 piece ('x')
# (this IS the plane index of the old OpenSpiel observation: [empty, o, x]).
# --------------------------------------------------------------------------
class ConnectFourGame(_Game):
    ROWS, COLS = 6, 7
    name = "connect_four"

    def num_distinct_actions(self):
        return self.COLS

    def information_state_normalized_vector_shape(self):
        return [3, self.ROWS, self.COLS]

    def max_game_length(self):
        return self.ROWS * self.COLS

    def new_initial_state(self):
        return ConnectFourState(self)

    def __str__(self):
        return "connect_four()"


class ConnectFourState:
    def __init__(self, game):
        self._game = game
        self._board = [[0] * game.COLS for _ in range(game.ROWS)]
        self._history = []
        self._outcome = None  # None = running, else returns()[0] in {-1,0,1}

    def clone(self):
        s = ConnectFourState.__new__(ConnectFourState)
        s._game = self._game
        s._board = [row[:] for row in self._board]
        s._history = self._history[:]
        s._outcome = self._outcome
        return s

    def current_player(self):
        if self._outcome is not None:
            return TERMINAL_PLAYER
        return len(self._history) & 1

    def is_terminal(self):
        return self._outcome is not None

    def history(self):
        return self._history[:]

    def legal_actions(self, player=None):
        if self._outcome is not None:
            return []
        top = self._game.ROWS - 1
        return [c for c in range(self._game.COLS) if self._board[top][c] == 0]

    def apply_action(self, action):
        g = self._game
        assert self._outcome is None
        action = int(action)
        player = len(self._history) & 1
        mark = 2 if player == 0 else 1
        row = 0
        while row < g.ROWS and self._board[row][action] != 0:
            row += 1
        assert row < g.ROWS, "illegal action %d" % action
        self._board[row][action] = mark
        self._history.append(action)
        if self._wins(row, action, mark):
            self._outcome = 1.0 if player == 0 else -1.0
        elif len(self._history) == g.ROWS * g.COLS:
            self._outcome = 0.0

    def _wins(self, r, c, mark):
        g = self._game
        for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
            n = 1
            for sgn in (1, -1):
                rr, cc = r + sgn * dr, c + sgn * dc
                while 0 <= rr < g.ROWS and 0 <= cc < g.COLS and self._board[rr][cc] == mark:
                    n += 1
                    rr += sgn * dr
                    cc += sgn * dc
            if n >= 4:
                return True
        return False

    def returns(self):
        z = 0.0 if self._outcome is None else self._outcome
        return [z, -z]

    def player_return(self, player):
        return self.returns()[player]

    def information_state(self, player=None):
        return ", ".join(str(a) for a in self._history)

    def information_state_as_normalized_vector(self, player=None):
        g = self._game
        n = g.ROWS * g.COLS
        v = [0.0] * (3 * n)
        for r in range(g.ROWS):
            for c in range(g.COLS):
                v[self._board[r][c] * n + r * g.COLS + c] = 1.0
        return v

    def __str__(self):
        ch = ".ox"
        return "\n".join("".join(ch[v] for v in row) for row in reversed(self._board))


# --------------------------------------------------------------------------
# breakthrough(rows=R, columns=C).  cell values: 0 black (player 0), 1 white
# (player 1), 2 empty — again the observation plane index.  Black starts on
# rows 0..1 and moves toward higher rows; white starts on rows R-2..R-1.
# action = ((r*C + c)*6 + dir)*2 + capture,
# dir -> (dr, dc): 0:(+1,-1) 1:(+1,0) 2:(+1,+1) 3:(-1,-1) 4:(-1,0) 5:(-1,+1)
# --------------------------------------------------------------------------
_BT_DIRS = ((1, -1), (1, 0), (1, 1), (-1, -1), (-1, 0), (-1, 1))


class BreakthroughGame(_Game):
    def __init__(self, rows=8, cols=8):
        self.ROWS, self.COLS = int(rows), int(cols)
        self.name = "breakthrough(rows=%d,columns=%d)" % (self.ROWS, self.COLS)

    def num_distinct_actions(self):
        return self.ROWS * self.COLS * 6 * 2

    def information_state_normalized_vector_shape(self):
        return [3, self.ROWS, self.COLS]

    def max_game_length(self):
        # every ply moves one piece one row forward; a piece may not reach the
        # far row without ending the game
        return 2 * self.COLS * (2 * self.ROWS - 5) + 1

    def new_initial_state(self):
        return BreakthroughState(self)

    def __str__(self):
        return self.name


class BreakthroughState:
    def __init__(self, game):
        self._game = game
        R, C = game.ROWS, game.COLS
        self._board = [[2] * C for _ in range(R)]
        for c in range(C):
            self._board[0][c] = 0
            self._board[1][c] = 0
            self._board[R - 2][c] = 1
            self._board[R - 1][c] = 1
        self._pieces = [2 * C, 2 * C]
        self._history = []
        self._winner = None

    def clone(self):
        s = BreakthroughState.__new__(BreakthroughState)
        s._game = self._game
        s._board = [row[:] for row in self._board]
        s._pieces = self._pieces[:]
        s._history = self._history[:]
        s._winner = self._winner
        return s

    def current_player(self):
        if self._winner is not None:
            return TERMINAL_PLAYER
        return len(self._history) & 1

    def is_terminal(self):
        return self._winner is not None

    def history(self):
        return self._history[:]

    def legal_actions(self, player=None):
        if self._winner is not None:
            return []
        g = self._game
        R, C = g.ROWS, g.COLS
        me = len(self._history) & 1
        dirs = (0, 1, 2) if me == 0 else (3, 4, 5)
        out = []
        for r in range(R):
            for c in range(C):
                if self._board[r][c] != me:
                    continue
                for d in dirs:
                    dr, dc = _BT_DIRS[d]
                    r2, c2 = r + dr, c + dc
                    if not (0 <= r2 < R and 0 <= c2 < C):
                        continue
                    tgt = self._board[r2][c2]
                    if tgt == 2:
                        out.append(((r * C + c) * 6 + d) * 2)
                    elif tgt == 1 - me and dc != 0:
                        out.append(((r * C + c) * 6 + d) * 2 + 1)
        return out  # ascending by construction

    def apply_action(self, action):
        g = self._game
        R, C = g.ROWS, g.COLS
        assert self._winner is None
        action = int(action)
        me = len(self._history) & 1
        cap = action & 1
        d = (action >> 1) % 6
        cell = (action >> 1) // 6
        r, c = divmod(cell, C)
        dr, dc = _BT_DIRS[d]
        r2, c2 = r + dr, c + dc
        assert self._board[r][c] == me, "no own piece on source"
        assert 0 <= r2 < R and 0 <= c2 < C
        tgt = self._board[r2][c2]
        if cap:
            assert tgt == 1 - me and dc != 0
            self._pieces[1 - me] -= 1
        else:
            assert tgt == 2
        self._board[r][c] = 2
        self._board[r2][c2] = me
        self._history.append(action)
        if (me == 0 and r2 == R - 1) or (me == 1 and r2 == 0) or self._pieces[1 - me] == 0:
            self._winner = me

    def returns(self):
        if self._winner is None:
            return [0.0, 0.0]
        return [1.0, -1.0] if self._winner == 0 else [-1.0, 1.0]

    def player_return(self, player):
        return self.returns()[player]

    def information_state(self, player=None):
        return ", ".join(str(a) for a in self._history)

    def information_state_as_normalized_vector(self, player=None):
        g = self._game
        n = g.ROWS * g.COLS
        v = [0.0] * (3 * n)
        for r in range(g.ROWS):
            for c in range(g.COLS):
                v[self._board[r][c] * n + r * g.COLS + c] = 1.0
        return v

    def __str__(self):
        ch = "bw."
        return "\n".join("".join(ch[v] for v in row) for row in self._board)


def load_game(name):
    """Parse the reference's game-name strings (train.py:24)."""
    name = name.strip()
    if name in ("connect_four", "connect_four()"):
        return ConnectFourGame()
    if name.startswith("breakthrough"):
        rows = cols = 8
        if "(" in name:
            args = name[name.index("(") + 1:name.rindex(")")]
            for kv in filter(None, (s.strip() for s in args.split(","))):
                k, v = kv.split("=")
                if k.strip() == "rows":
                    rows = int(v)
                elif k.strip() == "columns":
                    cols = int(v)
        return BreakthroughGame(rows, cols)
    raise ValueError("unknown game %r" % name)
