"""QLearner for VDN / QMIX / QPLEX (mirror of reference algorithm/q_learner.py:10-262).

``train(batch, train_step)`` keeps the reference contract (returns the float loss, syncs targets
every ``target_update_cycle``), but is an explicit forward/backward schedule of HIP kernels:
three agent unrolls, mixer forward (+target), TD loss, mixer backward, BPTT, fused clip+optimizer.
Reference quirks Q1 (double-Q pass continues from the eval net's final hidden state), Q2
(get_max_episode_len ignores unterminated episodes), Q4/Q5 (mask constant, first-index argmax) and
Q6 (target sync rule) are reproduced.
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch

from .. import ops
from ..hostutil import require_cuda, DeviceBatch, flatten_module
from ..rollout import EpisodeBatch
from ..network.mixer import VDNMixer, QMixMixer, DMAQer
from .common import (MASK_BIG, LearnerParams, FlatView, FusedOptimizer, Scratch, agent_backward, GradReducer, PairedUnroll, ResumeMixin, LossReadback, SpeculativeBatchMixin,
                     GraphedUpdate)


class QLearner(ResumeMixin, SpeculativeBatchMixin):
    def __init__(self, mac, args):
        self.max_episode_len = args.episode_limit
        self.gamma = args.gamma
        self.lr = args.lr
        self.model_dir = args.model_dir + '/' + args.alg + '/' + args.map
        self.args = args
        self.device = require_cuda("QLearner")

        self.eval_net = mac
        self.eval_net.cuda()
        self.target_net = copy.deepcopy(mac)
        if args.alg == 'vdn':
            self.mixer = VDNMixer(args)
        elif args.alg == 'qmix':
            self.mixer = QMixMixer(args)
        elif args.alg == 'qplex':
            self.mixer = DMAQer(args)
        else:
            raise ValueError("Mixer {} not recognised.".format(args.alg))
        self.target_mixer = copy.deepcopy(self.mixer)
        self.params = list(mac.parameters()) + list(self.mixer.parameters())
        self.cuda()

        self.optimizer = FusedOptimizer(self._flat, args.optimizer, self.lr, args.grad_norm_clip)
        self._buf = Scratch()
        self.reducer = GradReducer()
        from ..network import mixer as _mixer
        self.pair = PairedUnroll(x6=getattr(args, "gemm_mode", _mixer.DEFAULT_GEMM_MODE) == "bf16x6")
        self.loss_readback = LossReadback(args)
        self.graphs = GraphedUpdate.from_args(args)
        self.needs_avail = args.alg == 'qplex'       # the current-step availability masks the greedy action (:135-140)
        self.last_stats = None
        self.sync_replicas()

    def sync_replicas(self):
        """Data-parallel replicas start from rank 0's parameters, targets and optimizer state (called after
        construction and after load_models; a no-op without a process group)."""
        o = self.optimizer
        self.reducer.broadcast_(self._flat.flat, self.target_net.agent._flat.flat,
                                self.target_mixer._flat.flat if self.target_mixer._flat.n else None, o.s1, o.s2)

    # ------------------------------------------------------------------ storage
    def cuda(self):
        """Move everything to the MI355X and (re)build the flat buffers."""
        dev = self.device
        self.mixer.to(dev)
        self.target_mixer.to(dev)
        self.eval_net.agent.to(dev)
        self._flat = LearnerParams(self.params, dev)
        self.eval_net.agent._flat = FlatView(self._flat.flat, self.eval_net.agent.parameters(), 0)
        self.mixer._flat = FlatView(self._flat.flat, self.mixer.parameters(), self.eval_net.agent._flat.n)
        self.eval_net._dev = dev
        self.target_net._dev = dev
        self.target_net.agent.to(dev)
        flatten_module(self.target_net.agent, dev)
        flatten_module(self.target_mixer, dev)

    def _update_targets(self):
        """reference :181-184 - two device copies."""
        self.target_net.agent._flat.flat.copy_(self.eval_net.agent._flat.flat)
        if self.mixer is not None and self.mixer._flat.n:
            self.target_mixer._flat.flat.copy_(self.mixer._flat.flat)

    # ------------------------------------------------------------------ the hot path
    def get_max_episode_len(self, batch):
        """reference :49-66 (quirk Q2); returns the batch cut to [:, :T] and T."""
        T = DeviceBatch.first_terminated_len(torch.as_tensor(np.asarray(batch['terminated'])), self.args.episode_limit)
        for key in batch.keys():
            batch[key] = batch[key][:, :T]
        return batch, T

    def _forward_backward(self, db):
        a = self.args
        dev = self.device
        B, T, N, A, H = db.B, db.T, db.N, db.A, a.rnn_hidden_dim
        R, BT = B * T * N, B * T
        g = lambda name, shape, dt=torch.float32: self._buf.get(name, shape, dev, dt)
        # (no (B,T,N,H) hidden-state output: the Q-learning losses do not read it and BPTT finds h(t) in `saved`)
        q_evals, hs, saved = g("q_evals", (B, T, N, A)), None, g("saved", ops.saved_shape(T, B, N))
        h_last, h_scr = g("h_last", (B * N, H)), g("h_scr", (B * N, H))
        q_tgt, q_en = g("q_tgt", (B, T, N, A)), g("q_en", (B, T, N, A))
        q_chosen, q_tgt_chosen = g("q_chosen", (R,)), g("q_tgt_chosen", (R,))
        (oc, oc_bs, oc_t0), (on, on_bs, on_t0) = db.o_cur, db.o_next
        u_act = db.u_act.reshape(-1)

        # eval current-Q unroll (keeps activations), target next-Q unroll
        # (independent of each other: on small shards they run side by side on two streams, half of the CUs each)
        emap = getattr(db, 'o_map', None)
        # quirk Q1: no init_hidden between the two eval passes (reference :96-110) - the double-Q pass continues the eval chain.
        # Its inputs at steps 0..T-2 are the eval pass's inputs at steps 1..T-1 (same observations, same last actions, same
        # weights): fc1 and the input-side gate sums stored there are reused (gi), where the kernels of this shape can
        cont, gi = None, None
        if a.double_q:
            shifted = on is oc and on_bs == oc_bs and on_t0 == oc_t0 + 1
            split = self.pair.chain_split(B * N, T, a.obs_shape)
            from .. import experiments
            # (the continuation keeps reading the eval pass's input-side gate sums at every batch size: 0.55 ms at 4096 envs against
            # 0.97 ms for a plain unroll in the round-6 decomposition, csrc/agent_x6p.hip - which the TARGET unroll below runs on)
            if shifted and ((self.eval_net.unroll_x6(B, T, oc) and experiments.get("fwd_xs") != 0) or
                            ops.agent_unroll_reuse_supported(B, T, N, a.obs_shape, A, split[0] if split else 256)):
                gi = g("gi", ops.saved_shape(T, B, N, planes=3))
            cont = lambda cu: self.eval_net.unroll(on, on_bs, on_t0, db.u_fed, db.u_bs, 0, B, T, q_en, None, h_scr, None, h0=h_last,
                                                   ep_len=db.ep_len, ep_map=emap, cu_budget=cu, gi_in=gi)
        self.pair.run_chain(B * N, T, a.obs_shape,
                            lambda cu: self.eval_net.unroll(oc, oc_bs, oc_t0, db.u_fed, db.u_bs, -1, B, T, q_evals, hs, h_last, saved,
                                                            h0=None, ep_len=db.ep_len, ep_map=emap, cu_budget=cu, gi_out=gi),
                            cont,
                            lambda cu: self.target_net.unroll(on, on_bs, on_t0, db.u_fed, db.u_bs, 0, B, T, q_tgt, None, None, None,
                                                              h0=None, ep_len=db.ep_len, ep_map=emap, cu_budget=cu))
        ops.q_gather(q_evals, u_act, q_chosen, R, A)
        cur_max = None
        if a.double_q:
            cur_max = g("cur_max", (R,), torch.int32)
            ops.q_double_select(q_en, q_tgt, db.avail_next, MASK_BIG, q_tgt_chosen, cur_max, R, A)
        else:
            ops.q_masked_max(q_tgt, db.avail_next, MASK_BIG, q_tgt_chosen, None, R, A)

        ctx = {}
        qc, qtc = q_chosen.view(BT, N), q_tgt_chosen.view(BT, N)
        if a.alg == 'qplex':
            max_q = g("max_q", (R,))
            ops.q_masked_max(q_evals, db.avail, MASK_BIG, max_q, None, R, A)
            v_tot, a_tot = self.mixer.hip_forward(qc, db.s, BT, u_idx=db.u_taken.reshape(-1), max_q=max_q.view(BT, N), ctx=ctx)
            q_tot = g("q_tot", (BT,))
            ops.vec_add(v_tot, a_tot, q_tot, BT)
            if a.double_q:
                tgt_max = g("tgt_max", (R,))
                ops.q_masked_max(q_tgt, db.avail_next, MASK_BIG, tgt_max, None, R, A)
                vt, at = self.target_mixer.hip_forward(qtc, db.s_next, BT, u_idx=cur_max, max_q=tgt_max.view(BT, N), tag="t")
                q_tot_tgt = g("q_tot_tgt", (BT,))
                ops.vec_add(vt, at, q_tot_tgt, BT)
            else:
                q_tot_tgt, _ = self.target_mixer.hip_forward(qtc, db.s_next, BT, tag="t")
        else:
            fold = a.alg == 'qmix' and getattr(self.mixer, "loss_backward_fused", None) is not None and \
                self.mixer.loss_backward_fused(db.s) and not getattr(a, "no_loss_fold", False)
            q_tot = g("q_tot", (BT,)) if fold else self.mixer.hip_forward(qc, db.s, BT, ctx=ctx)
            q_tot_tgt = self.target_mixer.hip_forward(qtc, db.s_next, BT, tag="t")

        # TD loss (un-normalised numerator + sum(mask) land in the tail of the gradient buffer)
        self._flat.zero_grad()
        if a.alg != 'qplex' and fold:
            # fused QMIX: eval-mixer forward, TD loss and mixer backward are ONE launch (the backward recomputes q_tot anyway)
            dq_chosen = self.mixer.hip_loss_backward(qc, db.s, BT, q_tot_tgt, db.r, db.term, db.padded, self.gamma,
                                                     self._flat.stats[:2], q_tot=q_tot)
        else:
            dq_tot = g("dq_tot", (BT,))
            ops.td_loss(q_tot, q_tot_tgt, db.r, db.term, db.padded, self.gamma, dq_tot, self._flat.stats[:2], BT)
            # backward: mixer, gather, BPTT
            dq_chosen = self.mixer.hip_backward(ctx, dq_tot, BT)
        # the loss reaches q_evals only through the gather above: hand BPTT the sparse (action, gradient) pairs
        # instead of scattering them into a dense (B,T,N,A) tensor
        agent_backward(self.eval_net, db, "cur", saved, hs, None, None, self._buf,
                       dq_idx=u_act, dq_val=dq_chosen.reshape(-1).contiguous())
        self._dbg = dict(q_evals=q_evals, q_targets=q_tgt, q_tot=q_tot, q_tot_target=q_tot_tgt)

    def train(self, batch, train_step):
        if self.graphs is not None and isinstance(batch, EpisodeBatch) and batch.ring is not None and \
                self.graphs.run(self, batch.ring, batch.index):
            return self._finish_update(train_step)
        if isinstance(batch, DeviceBatch):
            db = batch
        elif isinstance(batch, EpisodeBatch) and batch.ring is not None:
            # replay sample: big arrays are read in place from the ring through the episode index
            prep = self.graphs.prepared if self.graphs is not None else None
            if prep is not None:             # the graph path already gathered the small arrays and agreed on T
                self.graphs.prepared = None
                db = DeviceBatch.from_record(batch.ring, self.args, T=prep[1], index=batch.index, small=prep[0])
            else:
                small = batch.ring.select_small(batch.index, avail_cur=self.needs_avail)
                db = self._device_batch(batch.ring, batch.index, small)
                if db is None:
                    return self._finish_update(train_step)
        elif isinstance(batch, EpisodeBatch) and batch.record is not None:
            db = self._device_batch(batch.record, None, None)
            if db is None:
                return self._finish_update(train_step)
        else:
            T = None
            if self.reducer.enabled:   # shards must agree on T
                T = DeviceBatch.first_terminated_len(torch.as_tensor(np.asarray(batch['terminated'])),
                                                     self.args.episode_limit, reducer=self.reducer)
            db = DeviceBatch.from_dict(batch, self.args, self.device, T=T)
        self.max_episode_len = db.T
        self._forward_backward(db)
        return self._finish_update(train_step)

    def _finish_update(self, train_step):
        """gradient all-reduce, clip + optimizer, target sync, loss readback (reference :168-179)"""
        self.reducer.allreduce_(self._flat.gradx)
        stats = self._flat.stats
        self.optimizer.step(den=stats[1:2])
        if train_step > 0 and train_step % self.args.target_update_cycle == 0:
            self._update_targets()
        self.last_stats = stats
        return self.loss_readback.read(stats[:2], lambda s: s[0] / s[1])

    # ------------------------------------------------------------------ checkpoints (reference :193-209)
    def save_models(self, train_step):
        num = str(train_step // self.args.save_cycle)
        if not os.path.exists(self.model_dir):
            os.makedirs(self.model_dir)
        self.eval_net.save_models(self.model_dir + '/' + num + '_rnn_net_params.pkl')
        torch.save({k: v.detach().cpu() for k, v in self.mixer.state_dict().items()},
                   self.model_dir + '/' + num + '_mixer_net_params.pkl')

    def load_models(self):
        if os.path.exists(self.model_dir + '/rnn_net_params.pkl'):
            path_rnn = self.model_dir + '/rnn_net_params.pkl'
            path_mix = self.model_dir + '/mixer_net_params.pkl'
            self.eval_net.load_models(path_rnn)
            self.mixer.load_state_dict(torch.load(path_mix, map_location='cpu'))
            self.sync_replicas()
            print('Successfully load the model: {} and {}'.format(path_rnn, path_mix))
        else:
            raise Exception("No model!")

    def get_q_and_q_tot_table(self):
        """Matrix-game diagnostic (reference :211-262): 3x3 q_tot table + per-agent Q rows with
        obs = state = 1 and zero last action (quirk Q9)."""
        one = {'o': np.ones((1, 1, 2, 1)), 's': np.ones((1, 1, 1)), 'o_next': np.ones((1, 1, 2, 1)),
               'u_onehot': np.zeros((1, 1, 2, 3)), 'avail_u': np.ones((1, 1, 2, 3))}
        self.eval_net.init_hidden(1)
        q_values, _ = self.eval_net.get_current_q_values(one, 1)     # (1,1,2,3)
        qv = q_values.cpu()
        q_table_i, q_table_j = qv[0, 0, 0].numpy(), qv[0, 0, 1].numpy()
        q_tot_table = np.zeros((3, 3))
        s = torch.ones(1, 1, 1)
        for i in range(3):
            for j in range(3):
                chosen = torch.stack((qv[:, :, 0, i], qv[:, :, 1, j]), dim=1).view(1, 1, 2)
                if self.args.alg == 'qplex':
                    v_tot = self.mixer(chosen, s, is_v=True)
                    onehot = torch.zeros(1, 1, 2, 3)
                    onehot[0, 0, 0, i] = 1
                    onehot[0, 0, 1, j] = 1
                    a_tot = self.mixer(chosen, s, actions=onehot, max_q_i=qv.max(dim=3)[0], is_v=False)
                    q_tot_table[i, j] = float(v_tot.item() + a_tot.item())
                else:
                    q_tot_table[i, j] = self.mixer(chosen, s).item()
        return q_tot_table, q_table_i, q_table_j
