"""TEST INFRASTRUCTURE — a deterministic stand-in for the PV-net.

`policy_fn(state) -> (priors[A], value)` is the seam the reference's search is
written against (/root/reference/mcts.py:146, examplegenerator.py:44-54).  For
parity tests the SAME function of the (C+1,H,W) board must feed the reference,
the C restatement and the HIP engine, so it is defined here on the flattened
0/1 board only, in integer arithmetic + one float32 softmax:

    h      = fnv1a64(board bytes) ^ salt
    logit_a= 4*u01(splitmix64(h + a*PHI)) - 2           (float64 -> float32)
    priors = softmax_f32(logits)      (float32, like Net.forward's F.softmax)
    value  = float32(2*u01(splitmix64(h ^ VSALT)) - 1)

The reference then sees `(priors.tolist(), float(value))`, exactly what
handle_gpu's `.tolist()` hands back (examplegenerator.py:74-77).
"""
import numpy as np

_M64 = (1 << 64) - 1
_PHI = 0x9E3779B97F4A7C15
_VSALT = 0xD1B54A32D192ED03


def _splitmix64(x):
    x = (x + _PHI) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _fnv1a64(data):
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def _u01(z):
    return (z >> 11) * (1.0 / 9007199254740992.0)


def board_key(board):
    """board: array-like (C+1,H,W) of 0/1 (+ player plane) -> bytes"""
    return np.asarray(board, dtype=np.float64).astype(np.uint8).tobytes()


def fake_eval(board, num_actions, salt=0):
    """-> (priors float32[A], value float32 scalar)"""
    h = _fnv1a64(board_key(board)) ^ (salt & _M64)
    logits = np.empty(num_actions, dtype=np.float64)
    for a in range(num_actions):
        logits[a] = 4.0 * _u01(_splitmix64((h + a * _PHI) & _M64)) - 2.0
    lg = logits.astype(np.float32)
    e = np.exp(lg - lg.max(), dtype=np.float32)
    pri = (e / e.sum(dtype=np.float32)).astype(np.float32)
    val = np.float32(2.0 * _u01(_splitmix64(h ^ _VSALT)) - 1.0)
    return pri, val


def fake_eval_batch(boards, num_actions, salt=0):
    pri = np.empty((len(boards), num_actions), dtype=np.float32)
    val = np.empty((len(boards),), dtype=np.float32)
    for i, b in enumerate(boards):
        pri[i], val[i] = fake_eval(b, num_actions, salt)
    return pri, val


def make_policy_fn(state_to_board, state_shape, num_actions, salt=0, log=None):
    """policy_fn for the reference / façade: state -> (list[A] of python floats, float)."""

    def policy_fn(state):
        board = state_to_board(state, state_shape)
        pri, val = fake_eval(board, num_actions, salt)
        if log is not None:
            log.append(board_key(board))
        return pri.tolist(), float(val)

    return policy_fn
