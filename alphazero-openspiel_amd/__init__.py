"""alphazero_openspiel_amd — MI355X-native AlphaZero self-play engine.

Drop-in for ONE path of danielwillemsen/alphazero-openspiel: mcts.py's PUCT search,
game_utils.play_game_self's rollout loop, alphazerobot.py's agent step and
examplegenerator.py's orchestration, for connect_four and breakthrough, producing the reference's
`[info_state, board, pi, z]` training examples.  The search runs as HIP kernels for gfx950 behind the
C ABI of include/az_engine.h; this package is the Python mirror of the reference's interface:

    from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
    from alphazero_openspiel_amd.alphazerobot import AlphaZeroBot
    from alphazero_openspiel_amd.network import Net, state_to_board
    from alphazero_openspiel_amd.game_utils import play_game_self
    from alphazero_openspiel_amd import games as pyspiel        # load_game(...)
"""
__version__ = "0.1.0"

from . import games  # noqa: F401  (pure host logic; importing it never touches the GPU)
