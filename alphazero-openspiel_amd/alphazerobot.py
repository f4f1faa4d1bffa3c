"""`AlphaZeroBot` with the reference's interface (alphazerobot.py:21-93) on the GPU search.

step(state) = re-root the device tree from state.history(), run the search in HIP kernels
(mcts.MCTS façade), then the reference's host arithmetic on the visit counts — mask + renormalise,
temperature, `np.random.choice` from numpy's global stream — so seeded runs draw what the reference
draws.  The bulk self-play path does NOT go through this class: ExampleGenerator keeps thousands of
games resident on the device and samples moves there (Philox).
"""
import numpy as np

from .mcts import MCTS


def remove_illegal_actions(action_probabilities, legal_actions):
    """Zero illegal entries, renormalise; all-illegal mass -> uniform over legal (alphazerobot.py:7-18)."""
    keep = np.zeros(action_probabilities.shape, dtype=bool)
    keep[legal_actions] = True
    action_probabilities[~keep] = 0.0
    total = np.sum(action_probabilities)
    if total > 1e-6:
        return action_probabilities / total
    uniform = np.zeros(len(action_probabilities))
    uniform[legal_actions] = 1. / len(legal_actions)
    return uniform


class AlphaZeroBot:
    def __init__(self, game, player, policy_fn, self_play=False, keep_search_tree=True, **kwargs):
        self.num_distinct_actions = game.num_distinct_actions()
        self.player = player
        self.policy_fn = policy_fn
        self.kwargs = kwargs
        self.use_probabilistic_actions = bool(self_play) or bool(kwargs.get("use_probabilistic_actions"))
        self.use_random_actions = bool(kwargs.get("use_random_actions", False))
        self.num_probabilistic_actions = int(kwargs.get("num_probabilistic_actions", 1000))
        self.temperature = float(kwargs.get("temperature", 1.0))
        self.self_play = self_play
        self.keep_search_tree = keep_search_tree
        self.mcts = MCTS(policy_fn, self.num_distinct_actions, **kwargs)

    def step(self, state):
        """-> (policy [(action, prob) for legal actions], action)"""
        if self.keep_search_tree:
            hist = state.history()
            if self.self_play:
                if hist:
                    self.mcts.update_root(hist[-1])
            elif len(hist) >= 2:
                self.mcts.update_root(hist[-2])
                self.mcts.update_root(hist[-1])
        else:
            old = self.mcts
            self.mcts = MCTS(self.policy_fn, self.num_distinct_actions, **self.kwargs)
            self.mcts._engine, self.mcts._io, self.mcts._evaluator = old._engine, old._io, old._evaluator
            if old._engine is not None:
                self.mcts._game = old._game
            self.mcts._history = None  # forces a fresh tree at the next search

        visits = np.array(self.mcts.search(state))
        legal = state.legal_actions(state.current_player())
        pi = remove_illegal_actions(visits, legal)
        tempered = pi ** (1. / self.temperature) / sum(pi ** (1. / self.temperature))
        n_played = len(state.history())
        if self.use_random_actions and n_played < self.num_probabilistic_actions:
            action = np.random.choice(legal)
        elif self.use_probabilistic_actions and n_played < self.num_probabilistic_actions:
            action = np.random.choice(len(tempered), p=tempered)
        else:
            action = np.argmax(tempered)
        return [(a, pi[a]) for a in legal], action
