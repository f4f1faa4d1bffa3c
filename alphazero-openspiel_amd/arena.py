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

_ENGINE_KW = ("n_playouts", "c_puct", "temperature", "keep_search_tree", "use_puct", "use_probabilistic_actions",
              "num_probabilistic_actions")


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


def run_duel(eng_a, eng_b, ev_a, ev_b, n_games, seed=None, check_every=16, use_graph=True, max_ticks=None):
    """Two arena engines facing each other (opponent="external", arena_flip False / True), each with its own evaluator:
    test_zero_vs_zero (game_utils.py:120-145).  Games = slots (no refill).  Returns (ret0 [n_games], progress of a)."""
    if n_games > eng_a.G or eng_a.G != eng_b.G:
        raise EngineError("a duel plays one game per slot: n_games <= n_slots, equal on both engines")
    eng_a.reset(n_games, seed)
    eng_b.reset(n_games, None if seed is None else seed + 1)
    io_a, io_b = eng_a.alloc_io(), eng_b.alloc_io()

    def tick():
        eng_a.advance(io_a[1], io_a[2], io_a[0])
        eng_b.advance(io_b[1], io_b[2], io_b[0])
        eng_a.exchange_moves(eng_b)
        ev_a(*io_a)
        ev_b(*io_b)

    ticks, graph = 0, None
    dev = eng_a.device
    if use_graph:
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                tick()
                ticks += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
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
        if eng_a.games_done() >= n_games and eng_b.games_done() >= n_games:
            break
        if max_ticks is not None and ticks >= max_ticks:
            raise EngineError("duel did not finish within %d ticks: %r / %r" % (max_ticks, eng_a.progress(), eng_b.progress()))
    prog = eng_a.progress()
    prog["ticks"] = ticks
    ex_a, ex_b = eng_a.export(), eng_b.export()
    if not ((ex_a["game_len"] == ex_b["game_len"]).all() and (ex_a["game_ret0"] == ex_b["game_ret0"]).all()):
        raise EngineError("the two engines of a duel disagree about the games they played")
    return ex_a["game_ret0"].copy(), prog, ex_a


def play_zero_vs_zero(policy_fn, policy_fn2, game_name, n_tests, settings1=None, settings2=None, device=None, seed=None,
                      eval_backend="fused", eval_precision="f32x", use_dirichlet=True):
    """n_tests `test_zero_vs_zero` calls (2 games each: network 1 first, network 1 second) in one batch: two AlphaZero agents
    with their own networks and settings (n_playouts, c_puct, ...), both with root noise as in the reference
    (game_utils.py:131-132).  -> (score1 [n_tests], score2 [n_tests], progress), scores from network 1's point of view."""
    from .mcts import _as_module
    if device is None:
        mod = _as_module(policy_fn)
        device = next(mod.parameters()).device if mod is not None and next(mod.parameters()).is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    n_games = 2 * int(n_tests)
    if n_games > 4096:
        raise ValueError("at most 2048 tests per call (one game per slot)")
    seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
    engines, evals = [], []
    try:
        for flip, (fn, settings) in enumerate(((policy_fn, settings1), (policy_fn2 if policy_fn2 is not None else policy_fn, settings2))):
            kw = {k: v for k, v in dict(settings or {}).items() if k in _ENGINE_KW + ("dirichlet_ratio",)}
            kw.setdefault("n_playouts", 100)
            engines.append(SelfPlayEngine(game_name, n_games, max_games=n_games, device=device, seed=seed + 17 * flip,
                                          use_dirichlet=use_dirichlet, arena_agent="zero", opponent="external", arena_flip=bool(flip), **kw))
            evals.append(_evaluator(fn, device, eval_backend, eval_precision, n_games))
        ret0, prog, _ = run_duel(engines[0], engines[1], evals[0], evals[1], n_games, seed=seed)
    finally:
        for e in engines:
            e.close()
        for ev in evals:
            if hasattr(ev, "close"):
                ev.close()
    s1, s2 = pair_scores(ret0)
    return s1, s2, prog


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
