"""`play_game_self` with the reference's interface (game_utils.py:148-206): ONE self-play game through
the AlphaZeroBot façade (search on the GPU, one engine slot).  Bulk generation should use
ExampleGenerator, which keeps thousands of games on the device; this entry exists for drop-in parity
(same arguments, same example records, same numpy random stream).

The evaluation pairings `test_zero_vs_mcts`, `test_net_vs_mcts`, `test_zero_vs_random`, `test_net_vs_random`,
`test_zero_vs_zero` (game_utils.py:53-145) keep their signatures and return shapes; each call plays its two games on the device arena
(alphazero_openspiel_amd.arena).  `ExampleGenerator(is_test=True).generate_tests` runs n of them as one batch."""
import numpy as np

from .alphazerobot import AlphaZeroBot
from .games import load_game
from .network import state_to_board


def _a0gb_value(root):
    """Off-policy / A0GB target: follow argmax(N + P | N > 0) to a leaf (game_utils.py:182-194)."""
    node, value, mult = root, None, 1.0
    while not node.is_leaf():
        value = node.Q
        score = {a: (c.N + c.P if c.N > 0 else -99.0) for a, c in node.children.items()}
        node = node.children[max(score, key=score.get)]
        mult *= -1.0
    if node.N > 0:
        value = node.Q
        mult *= -1.0
    return value * mult


def play_game(game, player1, player2, generate_statistics=False):
    """One game between two step()-bots, player1 moving first -> returns()[0]; with generate_statistics also
    {"player1": [{"root": Node}, ...], "player2": [...]}: both bots' search-tree roots after every move
    (game_utils.py:16-35).  Host loop over façade bots (one engine slot each) - the batched arena is the fast path."""
    statistics = {"player1": [], "player2": []}
    state = game.new_initial_state()
    while not state.is_terminal():
        mover = player1 if len(state.history()) % 2 == 0 else player2
        _, action = mover.step(state)
        state.apply_action(int(action))
        if generate_statistics:  # MCTS.root is a fresh host snapshot of the device tree: no deepcopy needed
            statistics["player1"].append({"root": player1.mcts.root})
            statistics["player2"].append({"root": player2.mcts.root})
    if generate_statistics:
        return state.returns()[0], statistics
    return state.returns()[0]


def play_game_self(policy_fn, game_name, **kwargs):
    game = load_game(game_name)
    state = game.new_initial_state()
    shape = game.information_state_normalized_vector_shape()
    A = game.num_distinct_actions()
    bot = AlphaZeroBot(game, 0, policy_fn, self_play=True, **kwargs)
    backup = str(kwargs.get("backup", "on-policy"))
    examples = []
    while not state.is_terminal():
        policy, action = bot.step(state)
        lookup = dict(policy)
        pi = [lookup.get(a, 0.0) for a in range(A)]
        record = [state.information_state(), state_to_board(state, shape), pi, None]
        if backup == "soft-Z":
            record[3] = -bot.mcts.root.Q
        elif backup == "A0C":
            record[3] = max(c.Q if c.N > 0 else -99.0 for c in bot.mcts.root.children.values())
        elif backup == "off-policy":
            record[3] = _a0gb_value(bot.mcts.root)
        if backup in ("on-policy", "soft-Z", "A0C", "off-policy"):
            examples.append(record)
        state.apply_action(action)
    if backup == "on-policy":
        z = state.returns()[0]
        for rec in examples:
            rec[3] = z
            z *= -1
    return examples


# ---------------------------------------------------------------------- evaluation pairings (game_utils.py:53-128)
def _one_test(policy_fn, game_name, agent, opponent, sims, kwargs):
    from . import arena
    s1, s2, _ = arena.play_tests(policy_fn, game_name, 1, agent, opponent, opponent_sims=sims, **kwargs)
    return float(s1[0]), float(s2[0])


def test_zero_vs_mcts(policy_fn, max_search_nodes, game_name, **kwargs):
    """AlphaZeroBot (no root noise) against MCTSBot(uct_c=1, max_search_nodes, RandomRolloutEvaluator(1)), once as first
    and once as second player -> (score1, score2, None) (game_utils.py:68-83)."""
    return _one_test(policy_fn, game_name, "zero", "uct", max_search_nodes, kwargs) + (None,)


def test_net_vs_mcts(policy_fn, max_search_nodes, game_name, **kwargs):
    """NeuralNetBot against the same MCTSBot (game_utils.py:86-101)."""
    return _one_test(policy_fn, game_name, "net", "uct", max_search_nodes, kwargs) + (None,)


def test_zero_vs_random(policy_fn, game_name="connect_four", **kwargs):
    """AlphaZeroBot against the uniform random bot (game_utils.py:53-65; the reference hard-codes connect_four)."""
    return _one_test(policy_fn, game_name, "zero", "random", 0, kwargs) + (None,)


def test_net_vs_random(policy_fn, game_name, **kwargs):
    """NeuralNetBot against the uniform random bot -> (score1, score2) (game_utils.py:104-117)."""
    return _one_test(policy_fn, game_name, "net", "random", 0, kwargs)


def test_zero_vs_zero(policy_fn, max_search_nodes, game_name, policy_fn2=None, generate_statistics=False, **kwargs):
    """Two AlphaZeroBots (root noise on) with their own networks and `settings1` / `settings2`, each once as first player
    -> (score1, score2, statistics) from the first network's point of view (game_utils.py:120-145).  `max_search_nodes` is
    unused, as in the reference.  With generate_statistics the two games run through `play_game` on façade bots (the
    reference's own loop, one device search per step) and `statistics` = {"game1": {...}, "game2": {...}} holds both
    players' tree roots after every move; without it the pair runs on the batched device arena."""
    if generate_statistics:
        game = load_game(game_name)
        fn2 = policy_fn2 if policy_fn2 else policy_fn
        settings1, settings2 = dict(kwargs.get("settings1") or {}), dict(kwargs.get("settings2") or {})
        statistics = {}
        bot1 = AlphaZeroBot(game, 0, policy_fn, use_dirichlet=True, **settings1)
        bot2 = AlphaZeroBot(game, 1, fn2, use_dirichlet=True, **settings2)
        score1, statistics["game1"] = play_game(game, bot1, bot2, generate_statistics=True)
        bot1 = AlphaZeroBot(game, 1, policy_fn, use_dirichlet=True, **settings1)
        bot2 = AlphaZeroBot(game, 0, fn2, use_dirichlet=True, **settings2)
        score2, st2 = play_game(game, bot2, bot1, generate_statistics=True)
        st2["player1"], st2["player2"] = st2["player2"], st2["player1"]  # keyed by network, not by seat
        statistics["game2"] = st2
        return score1, -score2, statistics
    from . import arena
    s1, s2, _ = arena.play_zero_vs_zero(policy_fn, policy_fn2, game_name, 1, settings1=kwargs.get("settings1"),
                                        settings2=kwargs.get("settings2"))
    return float(s1[0]), float(s2[0]), {}


for _f in (test_zero_vs_mcts, test_net_vs_mcts, test_zero_vs_random, test_net_vs_random, test_zero_vs_zero):
    _f.__test__ = False  # reference names; not pytest cases
