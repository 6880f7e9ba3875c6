"""RNNQNet mirror (reference network/q_network.py:6-21): fc1 -> ReLU -> GRUCell -> fc2.

The nn.Linear / nn.GRUCell members are parameter containers only (state_dict keys fc1.*, rnn.*,
fc2.* as in the reference); the arithmetic is the persistent HIP unroll kernel
(marl_amd/csrc/agent.hip) called with T = 1.
"""
import weakref

import torch
import torch.nn as nn

from .. import ops
from ..hostutil import require_cuda


# module -> (parameter objects, their data pointers, marl_agent_weights_t): the struct is rebuilt only when a
# parameter's storage moved (flat-buffer adoption, .to(), load_state_dict into new storage) - walking
# named_parameters() on every launch cost ~15 us, five times per update
_WEIGHTS = weakref.WeakKeyDictionary()


class RNNQNet(nn.Module):
    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.input_shape = input_shape
        self.fc1 = nn.Linear(input_shape, args.rnn_hidden_dim)
        self.rnn = nn.GRUCell(args.rnn_hidden_dim, args.rnn_hidden_dim)
        self.fc2 = nn.Linear(args.rnn_hidden_dim, args.n_actions)
        if args.rnn_hidden_dim != 64:
            raise ValueError("the gfx950 agent kernel is specialised for rnn_hidden_dim = 64 (wave64)")

    def weights(self):
        """marl_agent_weights_t over the current parameter storage."""
        c = _WEIGHTS.get(self)
        if c is not None:
            plist, ptrs, w = c
            if all(q.data_ptr() == o and q.is_cuda for q, o in zip(plist, ptrs)):
                return w
        p = dict(self.named_parameters())
        dev = p["fc1.weight"].device
        if dev.type != "cuda":
            dev = require_cuda("RNNQNet")
            self.to(dev)
            p = dict(self.named_parameters())
        for k, v in p.items():
            if not v.data.is_contiguous():
                v.data = v.data.contiguous()
        w = ops.agent_weights({k: v.data for k, v in p.items()})
        plist = list(p.values())
        _WEIGHTS[self] = (plist, [q.data_ptr() for q in plist], w)
        return w

    def forward(self, obs, hidden_state):
        """obs (rows, input_shape) already concatenated; hidden (rows, H) -> (q, h)."""
        dev = require_cuda("RNNQNet.forward")
        w = self.weights()
        x = obs.to(device=dev, dtype=torch.float32).contiguous()
        rows = x.shape[0]
        h_in = hidden_state.reshape(-1, self.args.rnn_hidden_dim).to(device=dev, dtype=torch.float32).contiguous()
        q = torch.empty(rows, self.args.n_actions, device=dev)
        h = torch.empty(rows, self.args.rnn_hidden_dim, device=dev)
        # rows act as "episodes" with one agent; the whole vector is the observation segment
        ops.agent_unroll_fwd(w, x, 1, 0, None, 0, 0, h_in, q, None, h, None, rows, 1, 1, self.input_shape,
                             self.args.n_actions, last_action=False, reuse_network=False)
        return q, h
