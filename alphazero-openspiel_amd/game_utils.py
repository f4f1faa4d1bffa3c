"""`play_game_self` with the reference's interface (game_utils.py:148-206): ONE self-play game through
the AlphaZeroBot façade (search on the GPU, one engine slot).  Bulk generation should use
ExampleGenerator, which keeps thousands of games on the device; this entry exists for drop-in parity
(same arguments, same example records, same numpy random stream)."""
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
