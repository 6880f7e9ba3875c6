"""ReplayBuffer (mirror of reference common/replaybuffer.py:5-80).

Two storage modes, chosen by what is stored first:
  * host   - the reference's 11 float64 numpy arrays (for plain dict episodes),
  * device - (T+1)-slot EpisodeRecord on HBM for episodes produced by the batched rollout:
             o/o_next, s/s_next and avail_u/avail_u_next are stored once.
Ring-index arithmetic (``_get_storage_idx``) and uniform sampling WITH replacement via
``np.random.randint`` are the reference's (quirk Q13).
"""
from __future__ import annotations

import threading

import numpy as np
import torch

from ..env.synthetic_smac import EpisodeRecord
from ..rollout import EpisodeBatch
from ..hostutil import h2d_async


class ReplayBuffer:
    def __init__(self, args):
        self.args = args
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        self.state_shape = args.state_shape
        self.obs_shape = args.obs_shape
        self.size = args.buffer_size
        self.episode_limit = args.episode_limit
        self.current_idx = 0
        self.current_size = 0
        self.buffers = None          # host mode: dict of numpy arrays (allocated lazily)
        self.record = None           # device mode: EpisodeRecord with `size` episodes
        self.lock = threading.Lock()

    def _alloc_host(self):
        T, N, O, S, A, n = self.episode_limit, self.n_agents, self.obs_shape, self.state_shape, self.n_actions, self.size
        e = np.empty
        self.buffers = {'o': e([n, T, N, O]), 'u': e([n, T, N, 1]), 's': e([n, T, S]), 'r': e([n, T, 1]),
                        'o_next': e([n, T, N, O]), 's_next': e([n, T, S]), 'avail_u': e([n, T, N, A]),
                        'avail_u_next': e([n, T, N, A]), 'u_onehot': e([n, T, N, A]), 'padded': e([n, T, 1]),
                        'terminated': e([n, T, 1])}

    def _peek_contiguous(self, inc):
        """first ring slot the next ``store_episode`` of ``inc`` episodes will use, or None when that
        store wraps around (mirrors ``_get_storage_idx`` without advancing)."""
        if self.current_idx + inc <= self.size:
            return self.current_idx
        if self.current_idx < self.size:
            return None
        return 0 if inc <= self.size else None

    def next_slot_record(self, E, T, N, O, S, A, device):
        """Zero-copy store: an EpisodeRecord VIEW over the ring slots the next store_episode(E episodes)
        will fill, so the rollout kernel writes the episodes in place and the store is index
        bookkeeping only.  None when the slots are not contiguous or the buffer is in host mode."""
        if self.buffers is not None:
            return None
        with self.lock:
            i0 = self._peek_contiguous(E)
            if i0 is None:
                return None
            if self.record is None:
                self.record = EpisodeRecord(self.size, T, N, O, S, A, device)
            r = self.record
            if (r.T, r.N, r.O, r.S, r.A) != (T, N, O, S, A):
                return None
            view = r.slice(i0, i0 + E)
            view.sink_slot = i0
            return view

    def store_episode(self, episode_batch):
        rec = getattr(episode_batch, "record", None)
        batch_size = rec.E if rec is not None else episode_batch['o'].shape[0]
        with self.lock:
            idxs = self._get_storage_idx(inc=batch_size)
            if rec is not None and self.buffers is None:
                if self.record is None:
                    self.record = EpisodeRecord(self.size, rec.T, rec.N, rec.O, rec.S, rec.A, rec.obs.device)
                first = int(np.atleast_1d(idxs)[0])
                if rec.obs.data_ptr() == self.record.obs[first].data_ptr() and getattr(rec, "sink_slot", -1) == first:
                    return                       # written in place by the rollout (next_slot_record)
                if getattr(rec, "sink_slot", None) is not None:
                    rec = rec.clone()            # a view of this ring at other slots: avoid overlapping copies
                idx_t = torch.as_tensor(np.atleast_1d(idxs), dtype=torch.long, device=rec.obs.device)
                rec.copy_into(self.record, idx_t)
                return
            if self.record is not None:
                raise ValueError("this ReplayBuffer holds device records; store EpisodeBatch objects")
            if self.buffers is None:
                self._alloc_host()
            src = episode_batch.numpy() if isinstance(episode_batch, EpisodeBatch) else episode_batch
            for k in self.buffers:
                v = src[k]
                self.buffers[k][idxs] = v.cpu().numpy() if isinstance(v, torch.Tensor) else v

    def sample(self, batch_size, exclude=None):
        """Uniform WITH replacement (reference :54-60, quirk Q13).  ``exclude`` = (first, count): ring slots a rollout
        in flight is writing (overlapped runner) - never sampled; the draw is the reference's single randint call over
        the remaining slots."""
        if exclude is not None and exclude[0] < self.current_size:
            lo, cnt = int(exclude[0]), int(min(exclude[1], self.current_size - exclude[0]))
            if self.current_size - cnt <= 0:
                raise ValueError("every stored episode is being overwritten: nothing to sample")
            idx = np.random.randint(0, self.current_size - cnt, batch_size)
            idx = np.where(idx >= lo, idx + cnt, idx)
        else:
            idx = np.random.randint(0, self.current_size, batch_size)
        if self.record is not None:
            idx_t = h2d_async(idx, self.record.obs.device, torch.long)   # pinned staging: no host stall behind the queue
            return EpisodeBatch(ring=self.record, index=idx_t)     # read in place by the learners (no gather copy)
        return {k: self.buffers[k][idx] for k in self.buffers}

    def _get_storage_idx(self, inc=None):
        inc = inc or 1
        if self.current_idx + inc <= self.size:
            idx = np.arange(self.current_idx, self.current_idx + inc)
            self.current_idx += inc
        elif self.current_idx < self.size:
            overflow = inc - (self.size - self.current_idx)
            idx = np.concatenate([np.arange(self.current_idx, self.size), np.arange(0, overflow)])
            self.current_idx = overflow
        else:
            idx = np.arange(0, inc)
            self.current_idx = inc
        self.current_size = min(self.size, self.current_size + inc)
        if inc == 1:
            idx = idx[0]
        return idx
