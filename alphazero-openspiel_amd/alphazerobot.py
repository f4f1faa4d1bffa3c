"""`AlphaZeroBot` with the reference's interface (alphazerobot.py:21-93), search on the GPU.

One `step(state)`:
  1. bring the device tree's root to `state` (re-root by the last move(s) when `keep_search_tree`, else a
     fresh tree)                                                          reference alphazerobot.py:54-68
  2. run the S sequential playouts in the HIP kernels (mcts.MCTS façade)   reference alphazerobot.py:71
  3. host arithmetic on the root visit fractions, with numpy exactly as the reference does it — illegal-move
     mask + renormalisation, temperature, then `np.random.choice` / argmax — so a seeded run draws the same
     random numbers as the reference                                      reference alphazerobot.py:72-93

The bulk self-play path does NOT go through this class: ExampleGenerator keeps thousands of games resident on
the device and samples moves there (Philox).
"""
import numpy as np

from .mcts import MCTS


def remove_illegal_actions(action_probabilities, legal_actions):
    """alphazerobot.py:7-18 — drop the mass on illegal actions and renormalise; when (almost) no mass is left
    on legal ones, fall back to uniform over them."""
    legal_mask = np.zeros(action_probabilities.shape, dtype=bool)
    legal_mask[legal_actions] = True
    action_probabilities[~legal_mask] = 0.0
    mass = np.sum(action_probabilities)
    if mass > 1e-6:
        return action_probabilities / mass
    fallback = np.zeros(len(action_probabilities))
    fallback[legal_actions] = 1. / len(legal_actions)
    return fallback


def _tempered(pi, temperature):
    powered = pi ** (1. / temperature)
    return powered / sum(powered)


class AlphaZeroBot:
    def __init__(self, game, player, policy_fn, self_play=False, keep_search_tree=True, **kwargs):
        self.num_distinct_actions = game.num_distinct_actions()
        self.player, self.policy_fn, self.kwargs = player, policy_fn, kwargs
        self.self_play, self.keep_search_tree = self_play, keep_search_tree
        self.use_probabilistic_actions = bool(self_play) or bool(kwargs.get("use_probabilistic_actions"))
        self.use_random_actions = bool(kwargs.get("use_random_actions", False))
        self.num_probabilistic_actions = int(kwargs.get("num_probabilistic_actions", 1000))
        self.temperature = float(kwargs.get("temperature", 1.0))
        self.mcts = MCTS(policy_fn, self.num_distinct_actions, **kwargs)

    def _advance_tree(self, history):
        if not self.keep_search_tree:  # a new search tree every step (the engine slot is reused, its tree is not)
            previous = self.mcts
            self.mcts = MCTS(self.policy_fn, self.num_distinct_actions, **self.kwargs)
            self.mcts.adopt_engine(previous)
            return
        recent = history[-1:] if self.self_play else (history[-2:] if len(history) >= 2 else [])
        for move in recent:
            self.mcts.update_root(move)

    def _pick(self, tempered, legal, n_played):
        early = n_played < self.num_probabilistic_actions
        if self.use_random_actions and early:
            return np.random.choice(legal)
        if self.use_probabilistic_actions and early:
            return np.random.choice(len(tempered), p=tempered)
        return np.argmax(tempered)

    def step(self, state):
        """-> (policy [(action, prob) for legal actions] — un-tempered, action)"""
        history = state.history()
        self._advance_tree(history)
        fractions = np.array(self.mcts.search(state))
        legal = state.legal_actions(state.current_player())
        pi = remove_illegal_actions(fractions, legal)
        action = self._pick(_tempered(pi, self.temperature), legal, len(history))
        return [(a, pi[a]) for a in legal], action
