"""Multi-agent controller with shared parameters (mirror of reference
controller/share_params.py:8-182, class SharedMAC).

Same constructor / method surface; the network evaluation is the persistent HIP unroll kernel.
Batched extensions (not in the reference): ``step_batch`` for lock-step rollouts and the
``unroll`` primitive the learners use.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from ..hostutil import require_cuda, to_dev, onehot_to_index, flatten_module
from ..network.q_network import RNNQNet


class SharedMAC:
    def __init__(self, args):
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        self.state_shape = args.state_shape
        self.obs_shape = args.obs_shape
        self.args = args
        input_shape = self._get_input_shape()
        self._build_agents(input_shape)
        self.hidden_states = None   # (n_episodes, n_agents, hidden_dim)
        self._dev = None

    # ------------------------------------------------------------------ plumbing
    def _get_input_shape(self):
        """reference share_params.py:114-123"""
        shape = self.obs_shape
        if self.args.last_action:
            shape += self.n_actions
        if self.args.reuse_network:
            shape += self.n_agents
        return shape

    def _build_agents(self, input_shape):
        self.agent = RNNQNet(input_shape, self.args)

    def device(self):
        if self._dev is None:
            self._dev = require_cuda("SharedMAC")
        p = next(self.agent.parameters())
        if p.device != self._dev:
            self.cuda()
        return self._dev

    def cuda(self):
        dev = require_cuda("SharedMAC")
        self._dev = dev
        if next(self.agent.parameters()).device != dev or not hasattr(self.agent, "_flat"):
            self.agent.to(dev)
            flatten_module(self.agent, dev)

    def parameters(self):
        return self.agent.parameters()

    def load_state(self, other_mac):
        src, dst = getattr(other_mac.agent, "_flat", None), getattr(self.agent, "_flat", None)
        if src is not None and dst is not None and src.n == dst.n and _is_flat(other_mac.agent) and _is_flat(self.agent):
            dst.flat.copy_(src.flat)       # one D2D copy
        else:
            self.agent.load_state_dict(other_mac.agent.state_dict())

    def save_models(self, path):
        torch.save({k: v.detach().cpu() for k, v in self.agent.state_dict().items()}, path)

    def load_models(self, path):
        self.agent.load_state_dict(torch.load(path, map_location="cpu"))

    def init_hidden(self, episode_num):
        """zeros (episodes, N, H) - reference :74-76 (here on the device)."""
        self.hidden_states = torch.zeros((episode_num, self.n_agents, self.args.rnn_hidden_dim), device=self.device())

    # ------------------------------------------------------------------ serial action choice
    def choose_action(self, obs, last_action, agent_num, avail_actions, epsilon, evaluate=False):
        """One agent, one env (reference :37-72), same numpy RNG draw order:
        one uniform per call, one choice only when exploring."""
        dev = self.device()
        N, A, O = self.n_agents, self.n_actions, self.obs_shape
        avail = np.asarray(avail_actions)
        avail_ind = np.nonzero(avail)[0]
        la = -1
        if self.args.last_action:
            nz = np.nonzero(np.asarray(last_action))[0]
            la = int(nz[0]) if nz.size else -1
        obs_t = to_dev(np.asarray(obs, dtype=np.float32).reshape(1, O), dev)
        # run row `agent_num` of a 1-episode batch: pad the agent axis so the id block matches
        obs_full = torch.zeros(1, 1, N, O, device=dev)
        obs_full[0, 0, agent_num] = obs_t[0]
        ufed = torch.full((1, 1, N), -1, dtype=torch.int32, device=dev)
        ufed[0, 0, agent_num] = la
        q = torch.empty(1, 1, N, A, device=dev)
        h_in = self.hidden_states.reshape(N, -1).contiguous()
        h_out = torch.empty_like(h_in)
        ops.agent_unroll_fwd(self.agent.weights(), obs_full, N, 0, ufed, N, 0, h_in, q, None, h_out, None,
                             1, 1, N, O, A, self.args.last_action, self.args.reuse_network)
        self.hidden_states[0, agent_num] = h_out[agent_num]
        q_value = q[0, 0, agent_num].cpu()
        q_value[torch.as_tensor(avail, dtype=torch.float32) == 0.0] = -float("inf")
        if np.random.uniform() < epsilon:
            return np.random.choice(avail_ind)
        return torch.argmax(q_value)

    # ------------------------------------------------------------------ batched primitives
    def unroll_x6(self, B, T, obs=None):
        """True when a T-step unroll over B episodes runs on the bf16x6 split kernel (args.gemm_mode = "bf16x6", a shape
        csrc/agent_x6.hip covers and - when given - an observation tensor on a 16-byte boundary: a misaligned view takes the
        fp32 kernels instead of failing in the entry point)."""
        from ..network import mixer as _mixer
        return (T >= 4 and getattr(self.args, "gemm_mode", _mixer.DEFAULT_GEMM_MODE) == "bf16x6"
                and (obs is None or obs.data_ptr() % 16 == 0)
                and ops.agent_unroll_x6_supported(B, T, self.n_agents, self.obs_shape, self.n_actions, self.args.last_action,
                                                  self.args.reuse_network))

    def unroll(self, obs, obs_bs, obs_t0, ufed, u_bs, u_t0, B, T, q, hs=None, h_last=None, saved=None, h0="state",
               ep_len=None, ep_map=None, cu_budget=0, gi_out=None, gi_in=None):
        """T-step unroll over B episodes starting from self.hidden_states (or zeros if None)."""
        dev = self.device()
        N, A, O = self.n_agents, self.n_actions, self.obs_shape
        if h0 == "state":
            h0 = None if self.hidden_states is None else self.hidden_states.reshape(B * N, -1).contiguous()
        if self.unroll_x6(B, T, obs):
            # opt-in: every product of the unroll as six bf16 MFMA products (csrc/agent_x6.hip); same `saved` layout, so the
            # fp32 BPTT kernel runs from it; gi_out / gi_in pair up because every unroll of these dimensions comes here
            ops.agent_unroll_fwd_x6(self.agent.weights(), obs, obs_bs, obs_t0, ufed, u_bs, u_t0, h0, q, hs, h_last, saved,
                                    B, T, N, O, A, self.args.last_action, self.args.reuse_network, ep_len=ep_len, ep_map=ep_map,
                                    cu_budget=cu_budget, gi_out=gi_out, gi_in=gi_in)
            return
        ops.agent_unroll_fwd(self.agent.weights(), obs, obs_bs, obs_t0, ufed, u_bs, u_t0, h0, q, hs, h_last, saved,
                             B, T, N, O, A, self.args.last_action, self.args.reuse_network, ep_len=ep_len, ep_map=ep_map,
                             cu_budget=cu_budget, gi_out=gi_out, gi_in=gi_in)

    def _batch_unroll(self, batch, T, which):
        dev = self.device()
        N, A, O, H = self.n_agents, self.n_actions, self.obs_shape, self.args.rnn_hidden_dim
        key = "o" if which == "cur" else "o_next"
        obs = to_dev(batch[key][:, :T], dev)
        B = obs.shape[0]
        if "u_idx" in batch:
            uidx = to_dev(batch["u_idx"][:, :T], dev, torch.int32).view(B, T, N)
        else:
            uidx = onehot_to_index(to_dev(batch["u_onehot"][:, :T], dev)).view(B, T, N)
        q = torch.empty(B, T, N, A, device=dev)
        hs = torch.empty(B, T, N, H, device=dev)
        h_last = torch.empty(B * N, H, device=dev)
        self.unroll(obs, T * N, 0, uidx, T * N, -1 if which == "cur" else 0, B, T, q, hs, h_last)
        self.hidden_states = h_last.view(B, N, H)
        return q, hs

    def get_current_q_values(self, batch, max_episode_len):
        """q (B,T,N,A) and the hidden state after each step (B,T,N,H) - reference :125-146."""
        return self._batch_unroll(batch, max_episode_len, "cur")

    def get_next_q_values(self, batch, max_episode_len):
        """reference :148-168 (inputs o_next, u_onehot[t])."""
        return self._batch_unroll(batch, max_episode_len, "next")


def _is_flat(module):
    fp = getattr(module, "_flat", None)
    if fp is None:
        return False
    ps = list(module.parameters())
    return len(ps) == len(fp.params) and all(p.data.data_ptr() == fp.flat.data_ptr() + 4 * o
                                              for p, o in zip(ps, fp.offsets))


class _Unsupported:
    def __init__(self, *a, **k):
        raise NotImplementedError("%s is outside the MI355X hot path (see DESIGN.md, out of scope)" % type(self).__name__)


class SeparatedMAC(_Unsupported):
    """name kept for `from controller.share_params import ...` (reference runner.py:4); the reference class is broken."""


class SharedMACWithState(_Unsupported):
    pass


class RTWMAC(_Unsupported):
    pass
