"""TEST INFRASTRUCTURE — loader for the REAL reference implementation.

Works only where /root/reference is mounted (the build container).  It is used
by oracle/gen_golden.py to produce the committed fixtures under tests/golden/
and by the container-only tests that check the C restatement against the live
reference.  Nothing here travels to the GPU box in executable form other than
this loader itself; the reference's files are never copied.

Mechanism: OpenSpiel (`pyspiel`) is absent, so a stub module
exposing only the SYMBOLS the reference touches at import time (`Bot`, `Game`,
`load_game`, `open_spiel.python.algorithms.mcts`) is placed in sys.modules and
game dynamics come from oracle/pygames.py through the duck-typed state
protocol the reference itself sanctions (alphazerobot.py:27-28,
toy_domain.py:15-86).
"""
import importlib
import os
import sys
import types

REFERENCE_DIR = os.environ.get("AZ_REFERENCE_DIR", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "mcts.py"))


def _install_stub_pyspiel():
    from . import pygames

    if "pyspiel" in sys.modules and getattr(sys.modules["pyspiel"], "_az_stub", False):
        return sys.modules["pyspiel"]
    ps = types.ModuleType("pyspiel")
    ps._az_stub = True

    class Game:  # only used for `type(game) is pyspiel.Game`
        pass

    class Bot:
        def __init__(self, *a, **k):
            pass

    ps.Game = Game
    ps.Bot = Bot
    ps.load_game = pygames.load_game
    sys.modules["pyspiel"] = ps
    # `from open_spiel.python.algorithms import mcts` (game_utils.py:3) — arena only
    for name in ("open_spiel", "open_spiel.python", "open_spiel.python.algorithms",
                 "open_spiel.python.algorithms.mcts"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["open_spiel"].python = sys.modules["open_spiel.python"]
    sys.modules["open_spiel.python"].algorithms = sys.modules["open_spiel.python.algorithms"]
    sys.modules["open_spiel.python.algorithms"].mcts = sys.modules["open_spiel.python.algorithms.mcts"]
    return ps


class _Ref:
    pass


_cached = None


def load_reference():
    """Import the reference's hot-path modules unmodified; returns a namespace
    with .mcts .alphazerobot .game_utils .network .examplegenerator .train attributes (train.Trainer is only ever used
    through its static / unbound methods: instantiating it would open log files)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_DIR)
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
    _install_stub_pyspiel()
    saved = {}
    names = ("mcts", "network", "alphazerobot", "game_utils", "examplegenerator", "train")
    for n in names:  # do not clobber same-named modules of the caller
        if n in sys.modules:
            saved[n] = sys.modules.pop(n)
    sys.path.insert(0, REFERENCE_DIR)
    try:
        ref = _Ref()
        for n in names:
            setattr(ref, n, importlib.import_module(n))
    finally:
        sys.path.remove(REFERENCE_DIR)
    # keep them importable under private names only
    for n in names:
        sys.modules["_azref_" + n] = sys.modules.pop(n)
    sys.modules.update(saved)
    _cached = ref
    return ref
