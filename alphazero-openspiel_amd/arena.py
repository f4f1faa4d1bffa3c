"""Evaluation arena on the device: the network-driven agent against a bot, thousands of games at a time.

Reference: `game_utils.play_game` and the `test_*_vs_*` pairings (game_utils.py:16-145), driven in bulk by
`ExampleGenerator(is_test=True).generate_tests` (examplegenerator.py:25-37,177-195) from `Trainer.test_agent`
(train.py:238-270).  There every test is a worker process playing two games (agent first, agent second) with a pipe
round trip per leaf; here all games of an evaluation are slots of one engine:

  agent "zero"    AlphaZeroBot outside self-play (alphazerobot.py:42-93): n_playouts of PUCT search per move, no root
                  noise, the most visited move, tree kept across both players' moves      - az_advance_kernel
  agent "net"     NeuralNetBot (alphazerobot.py:96-120): argmax of the masked, renormalised priors
  opponent "uct"  open_spiel MCTSBot(uct_c, max_search_nodes, RandomRolloutEvaluator(1))   - az_opponent_kernel
  opponent "random"  pyspiel.make_uniform_random_bot

One arena tick = [az_engine_advance, az_engine_opponent_moves, PV-net forward].  Game id i gives the agent side i & 1:
the pair (2k, 2k+1) is one reference `test_*` call and scores  ret0(2k) - ret0(2k+1)  (score1 + score2).
"""
import numpy as np
import torch

from .engine import DeviceEvaluator, EngineError, SelfPlayEngine

_ENGINE_KW = ("n_playouts", "c_puct", "temperature", "keep_search_tree")


def arena_engine(game, n_slots, n_games, agent, opponent, opponent_sims=0, device=0, seed=0, **kwargs):
    """An engine configured for evaluation games.  kwargs: the AlphaZeroBot keywords the reference passes through
    (n_playouts default 100: mcts.py:98; c_puct, temperature; dirichlet_ratio is accepted and unused: the test pairings
    construct the bot with use_dirichlet=False, game_utils.py:72)."""
    kw = {k: kwargs[k] for k in _ENGINE_KW if k in kwargs}
    if agent == "net":
        kw.update(n_playouts=1, keep_search_tree=False)
    else:
        kw.setdefault("n_playouts", 100)
    return SelfPlayEngine(game, n_slots, max_games=n_games, device=device, seed=seed, use_dirichlet=False,
                          arena_agent=agent, opponent=opponent, opponent_sims=opponent_sims,
                          opponent_uct_c=float(kwargs.get("opponent_uct_c", 1.0)), **kw)


def run_arena(engine, evaluator, n_games, seed=None, check_every=16, use_graph=True, max_ticks=None):
    """Play n_games evaluation games (ids 0..n_games-1; the agent has side id & 1).  Returns (ret0 [n_games] float32 =
    returns()[0] of every game, progress dict)."""
    engine.reset(n_games, seed)
    obs, pri, val = engine.alloc_io()

    def tick():
        engine.advance(pri, val, obs)
        engine.opponent_moves()
        evaluator(obs, pri, val)

    ticks, graph = 0, None
    if use_graph:
        torch.cuda.synchronize(engine.device)
        side = torch.cuda.Stream(engine.device)
        side.wait_stream(torch.cuda.current_stream(engine.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                tick()
                ticks += 1
        torch.cuda.current_stream(engine.device).wait_stream(side)
        torch.cuda.synchronize(engine.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tick()
    while True:
        for _ in range(check_every):
            if graph is not None:
                graph.replay()
            else:
                tick()
            ticks += 1
        if engine.games_done() >= n_games:
            break
        if max_ticks is not None and ticks >= max_ticks:
            raise EngineError("arena did not finish within %d ticks: %r" % (max_ticks, engine.progress()))
    prog = engine.progress()
    prog["ticks"] = ticks
    ex = engine.export()
    return ex["game_ret0"].copy(), prog, ex


def pair_scores(ret0):
    """ret0 of games (2k, 2k+1) -> (score1, score2) per pair as the reference's test functions return them: score1 = the
    agent's result as first player, score2 = minus the first player's result when the agent is second (game_utils.py:76-82)."""
    r = np.asarray(ret0, dtype=np.float64)
    return r[0::2], -r[1::2]


def _evaluator(net_or_fn, device, backend, precision, n_slots):
    from .mcts import _as_module
    mod = _as_module(net_or_fn)
    if mod is None:
        raise TypeError("the device arena needs the network itself (an nn.Module or its bound .predict), not an arbitrary "
                        "python policy_fn")
    if backend == "fused":
        from .fusednet import FusedNet
        return FusedNet(mod, device, max_boards=n_slots, precision=precision)
    return DeviceEvaluator(mod, device)


def play_tests(policy_fn, game_name, n_tests, agent, opponent, opponent_sims=0, device=None, seed=None, n_slots=None,
               eval_backend="fused", eval_precision="f32x", **kwargs):
    """n_tests reference `test_*` calls (2 * n_tests games) in one batch -> (score1 [n_tests], score2 [n_tests], progress)."""
    from .mcts import _as_module
    if device is None:
        mod = _as_module(policy_fn)
        device = next(mod.parameters()).device if mod is not None and next(mod.parameters()).is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    n_games = 2 * int(n_tests)
    n_slots = int(n_slots or min(n_games, 4096))
    seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
    eng = arena_engine(game_name, n_slots, n_games, agent, opponent, opponent_sims, device=device, seed=seed, **kwargs)
    ev = _evaluator(policy_fn, device, eval_backend, eval_precision, n_slots)
    try:
        ret0, prog, _ = run_arena(eng, ev, n_games)
    finally:
        eng.close()
        if hasattr(ev, "close"):
            ev.close()
    s1, s2 = pair_scores(ret0)
    return s1, s2, prog
