"""PV-network under PyTorch-ROCm, checkpoint-compatible with the reference's `network.py`.

Same computation and the same `state_dict` keys as reference network.py:21-104 — a tower of
pre-activation residual blocks (BN -> LeakyReLU(0.01) -> 3x3 conv, twice; identity skip, or a 1x1 conv
skip when the channel count changes), one Linear from the flattened NCHW tower to A+1 logits, softmax
over the first A, tanh on the last — so `Net(shape, A).load_state_dict(torch.load(reference.pth))`
works.  The reference hard-codes 5 blocks x 50 filters (network.py:37-43); here both are
parameters (BASELINE configs name 2/10/20-block towers) with (5, 50) the default.

`state_to_board` is the host twin of the device observation writer (csrc/az_games.h az_obs_elem).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def state_to_board(state, state_shape):
    """(C+1,H,W) float64: C observation planes + a plane holding current_player() (network.py:9-18)."""
    c, h, w = state_shape
    board = np.empty((c + 1, h, w), dtype=np.float64)
    board[:c] = np.asarray(state.information_state_as_normalized_vector(), dtype=np.float64).reshape(c, h, w)
    board[c] = float(state.current_player())
    return board


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.bn1 = nn.BatchNorm2d(in_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.use_1x1conv = in_channels != out_channels
        if self.use_1x1conv:
            self.conv3 = nn.Conv2d(in_channels, out_channels, 1)

    def forward(self, x):
        y = self.conv1(F.leaky_relu(self.bn1(x)))
        y = self.conv2(F.leaky_relu(self.bn2(y)))
        return (self.conv3(x) if self.use_1x1conv else x) + y


class Net(nn.Module):
    def __init__(self, state_shape, num_distinct_actions, n_blocks=5, n_filters=50, **kwargs):
        super().__init__()
        self.state_shape = list(state_shape)
        self.num_filters_input = state_shape[0] + 1
        self.height, self.width = state_shape[1], state_shape[2]
        self.num_distinct_actions = num_distinct_actions
        self.n_blocks, self.n_filts = int(n_blocks), int(n_filters)
        self.device = kwargs.get("device", torch.device("cpu"))
        for i in range(self.n_blocks):
            setattr(self, "resblock%d" % (i + 1),
                    ResidualBlock(self.num_filters_input if i == 0 else self.n_filts, self.n_filts))
        self.fc1 = nn.Linear(self.n_filts * self.height * self.width, num_distinct_actions + 1)

    def blocks(self):
        return [getattr(self, "resblock%d" % (i + 1)) for i in range(self.n_blocks)]

    def forward(self, x):
        for blk in self.blocks():
            x = blk(x)
        x = self.fc1(x.reshape(-1, self.n_filts * self.height * self.width))
        logits, v = x.split(self.num_distinct_actions, dim=1)
        return F.softmax(logits, dim=1), torch.tanh(v)

    def predict(self, state):
        """policy_fn(state) -> (list[A], float)   (network.py:66-80)"""
        board = state_to_board(state, self.state_shape)
        dev = next(self.parameters()).device
        with torch.no_grad():
            p, v = self.forward(torch.from_numpy(board).float().unsqueeze(0).to(dev))
        return p[0].tolist(), float(v)


def net_from_state_dict(sd, state_shape, num_distinct_actions):
    """Build a Net whose depth/width are read off a reference-format state_dict (torch tensors or numpy)."""
    sd = {k: (torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v) for k, v in sd.items()}
    n_blocks = 1 + max(int(k.split(".")[0][len("resblock"):]) for k in sd if k.startswith("resblock")) - 1
    n_filters = sd["resblock1.conv1.weight"].shape[0]
    net = Net(state_shape, num_distinct_actions, n_blocks=n_blocks, n_filters=n_filters)
    net.load_state_dict(sd)
    return net.eval()  # inference use: BatchNorm on running statistics (train.py:89,154)


def load_npz_checkpoint(path, state_shape, num_distinct_actions):
    with np.load(path) as z:
        return net_from_state_dict({k: z[k] for k in z.files}, state_shape, num_distinct_actions)
