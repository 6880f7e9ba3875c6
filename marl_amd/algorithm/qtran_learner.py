"""QTRAN-base learner (mirror of reference algorithm/qtran_learner.py:10-272).

loss = L_td + lambda_opt * L_opt + lambda_nopt * L_nopt (reference :116-152) with the joint-Q,
target joint-Q and V heads of network/mixer.py on HIP kernels; gradients reach the agent both
through the individual Qs and through the per-step hidden states (dhs input of the BPTT kernel).
The reference's unused QMixMixer (``q_sum_mixer``, quirk Q11) is kept in the parameter list so the
optimizer state layout matches; its gradient is identically zero.
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch

from .. import ops
from ..hostutil import require_cuda, DeviceBatch, flatten_module
from ..rollout import EpisodeBatch
from ..network.mixer import QtranQBase, QtranQAlt, QtranV, QMixMixer
from .common import (MASK_BIG, MASK_QTRAN_EVAL, LearnerParams, FlatView, FusedOptimizer, Scratch, agent_backward,
                     GradReducer, PairedUnroll, ResumeMixin, LossReadback, SpeculativeBatchMixin, GraphedUpdate)


class QTRANLearner(ResumeMixin, SpeculativeBatchMixin):
    def __init__(self, mac, args):
        self.max_episode_len = args.episode_limit
        self.gamma = args.gamma
        self.lr = args.lr
        self.model_dir = args.model_dir + '/' + args.alg + '/' + args.map
        self.args = args
        self.device = require_cuda("QTRANLearner")

        self.eval_net = mac
        self.eval_net.cuda()
        self.target_net = copy.deepcopy(mac)
        if args.alg == 'qtran_base':
            self.mixer = QtranQBase(args)
        elif args.alg == 'qtran_alt':
            self.mixer = QtranQAlt(args)
        else:
            raise ValueError("Mixer {} not recognised.".format(args.alg))
        self.target_mixer = copy.deepcopy(self.mixer)
        self.params = list(mac.parameters()) + list(self.mixer.parameters())
        self.v = QtranV(args)
        self.params += list(self.v.parameters())
        self.q_sum_mixer = QMixMixer(args)
        self.params += list(self.q_sum_mixer.parameters())
        self.cuda()
        self.optimizer = FusedOptimizer(self._flat, args.optimizer, self.lr, args.grad_norm_clip)
        self._buf = Scratch()
        self.reducer = GradReducer()
        from ..network import mixer as _mixer
        self.pair = PairedUnroll(x6=getattr(args, "gemm_mode", _mixer.DEFAULT_GEMM_MODE) == "bf16x6")
        self.loss_readback = LossReadback(args)
        self.graphs = GraphedUpdate.from_args(args)
        self.needs_avail = True                      # local greedy actions are masked with the current availability (:103-108)
        self.last_stats = None
        self.sync_replicas()

    def sync_replicas(self):
        """replicas start from rank 0's parameters, targets and optimizer state (see QLearner.sync_replicas)"""
        o = self.optimizer
        self.reducer.broadcast_(self._flat.flat, self.target_net.agent._flat.flat, self.target_mixer._flat.flat, o.s1, o.s2)

    def cuda(self):
        dev = self.device
        for m in (self.mixer, self.target_mixer, self.v, self.q_sum_mixer, self.eval_net.agent, self.target_net.agent):
            m.to(dev)
        self._flat = LearnerParams(self.params, dev)
        off = 0
        for m in (self.eval_net.agent, self.mixer, self.v, self.q_sum_mixer):
            m._flat = FlatView(self._flat.flat, m.parameters(), off)
            off += m._flat.n
        self.eval_net._dev = dev
        self.target_net._dev = dev
        flatten_module(self.target_net.agent, dev)
        flatten_module(self.target_mixer, dev)

    def _update_targets(self):
        self.target_net.agent._flat.flat.copy_(self.eval_net.agent._flat.flat)
        self.target_mixer._flat.flat.copy_(self.mixer._flat.flat)

    def get_max_episode_len(self, batch):
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
        q_evals, hs, saved = g("q_evals", (B, T, N, A)), g("hs", (B, T, N, H)), g("saved", ops.saved_shape(T, B, N))
        q_tgt, hs_tgt = g("q_tgt", (B, T, N, A)), g("hs_tgt", (B, T, N, H))
        (oc, oc_bs, oc_t0), (on, on_bs, on_t0) = db.o_cur, db.o_next
        u_act = db.u_act.reshape(-1)
        u_taken = db.u_taken.reshape(-1)          # one-hot(u) with zeros on padding (batch['u_onehot'])

        emap = getattr(db, 'o_map', None)
        self.pair.run(B * N, T,
                      lambda cu: self.eval_net.unroll(oc, oc_bs, oc_t0, db.u_fed, db.u_bs, -1, B, T, q_evals, hs, None, saved,
                                                   h0=None, ep_len=db.ep_len, ep_map=emap, cu_budget=cu),
                      lambda cu: self.target_net.unroll(on, on_bs, on_t0, db.u_fed, db.u_bs, 0, B, T, q_tgt, hs_tgt, None, None,
                                                     h0=None, ep_len=db.ep_len, ep_map=emap, cu_budget=cu))

        # local greedy actions (reference :103-114): eval clone masked with -999999, targets with -9999999
        opt_eval, opt_tgt = g("opt_eval", (R,), torch.int32), g("opt_tgt", (R,), torch.int32)
        q_max_eval = g("q_max_eval", (R,))
        ops.q_masked_max(q_evals, db.avail, MASK_QTRAN_EVAL, q_max_eval, opt_eval, R, A)
        ops.q_masked_max(q_tgt, db.avail_next, MASK_BIG, None, opt_tgt, R, A)

        hs2, hst2 = hs.view(R, H), hs_tgt.view(R, H)
        ctx_q, ctx_v = {}, {}
        # the state columns of the joint-Q head's first layer are shared by the two evaluations of the eval mixer
        # and the V head reads the same states: both state parts come from one pass over s
        sp_e = sp_v = None
        if self.mixer._qt_ok(hs2):
            if self.v._qt_ok(hs2):
                sp_e, sp_v = self.mixer.state_part(db.s, BT, "e", other=self.v)
            else:
                sp_e = self.mixer.state_part(db.s, BT, "e")
        # the taken and the greedy actions of the eval mixer (:116, :133; the latter detached in the loss) in one launch
        joint_q, joint_q_hat = self.mixer.hip_forward(db.s, hs2, u_taken, BT, ctx=ctx_q, tag="e", sp=sp_e, u_idx2=opt_eval)
        joint_q_tgt = self.target_mixer.hip_forward(db.s_next, hst2, opt_tgt, BT, tag="t")
        v = self.v.hip_forward(db.s, hs2, BT, ctx=ctx_v, sp=sp_v)

        q_sum_opt, q_sum_nopt, q_ind = g("q_sum_opt", (BT,)), g("q_sum_nopt", (BT,)), g("q_ind", (R,))
        ops.agent_sum(q_max_eval, q_sum_opt, BT, N, 1)
        ops.q_gather(q_evals, u_act, q_ind, R, A)
        ops.agent_sum(q_ind, q_sum_nopt, BT, N, 1)

        self._flat.zero_grad()
        d_jq, d_v, d_so, d_sn = g("d_jq", (BT,)), g("d_v", (BT,)), g("d_so", (BT,)), g("d_sn", (BT,))
        ops.qtran_loss(joint_q, joint_q_tgt, v, joint_q_hat, q_sum_opt, q_sum_nopt, db.r, db.term, db.padded,
                       self.gamma, a.lambda_opt, a.lambda_nopt, d_jq, d_v, d_so, d_sn, self._flat.stats, BT)

        # backward: heads -> dhs, individual Qs -> dq, then BPTT
        dhs = g("dhs", (B, T, N, H))
        self.mixer.hip_backward(ctx_q, d_jq, BT, dhs.view(R, H), accumulate=False)
        self.v.hip_backward(ctx_v, d_v, BT, dhs.view(R, H), accumulate=True)
        # (the heads' row-level weight gradients on a side stream beside the agent's backward pass were measured: 300 -> 284-293
        # updates/s on the 512-env shard - the persistent BPTT workgroups hold every CU's LDS, the small launches only delay them)
        # the losses reach q_evals through two gathers per row - the taken action (L_nopt) and the greedy action (L_opt),
        # each with one gradient per (episode, step) shared by its agents: BPTT takes the two sparse (action, gradient)
        # pairs and the dense (B,T,N,A) gradient is never materialised
        agent_backward(self.eval_net, db, "cur", saved, hs, None, dhs, self._buf, dq_idx=u_act, dq_val=d_sn,
                       dq_idx2=opt_eval, dq_val2=d_so, dq_gdiv=N)
        self._dbg = dict(q_evals=q_evals, hs=hs, joint_q=joint_q, joint_q_targets=joint_q_tgt, v=v,
                         joint_q_hat=joint_q_hat)

    def train(self, batch, train_step):
        if self.graphs is not None and isinstance(batch, EpisodeBatch) and batch.ring is not None and \
                self.graphs.run(self, batch.ring, batch.index):
            db = None                    # forward / backward done (hipGraph replay on the static buffers)
        elif isinstance(batch, DeviceBatch):
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
        elif isinstance(batch, EpisodeBatch) and batch.record is not None:
            db = self._device_batch(batch.record, None, None)
        else:
            T = None
            if self.reducer.enabled:
                T = DeviceBatch.first_terminated_len(torch.as_tensor(np.asarray(batch['terminated'])),
                                                     self.args.episode_limit, reducer=self.reducer)
            db = DeviceBatch.from_dict(batch, self.args, self.device, T=T)
        if db is not None:               # (None: _device_batch already launched the pass for the record's full length)
            self.max_episode_len = db.T
            self._forward_backward(db)
        self.reducer.allreduce_(self._flat.gradx)
        st = self._flat.stats
        self.optimizer.step(den=st[3:4])
        if train_step > 0 and train_step % self.args.target_update_cycle == 0:
            self._update_targets()
        self.last_stats = st
        lo, ln = self.args.lambda_opt, self.args.lambda_nopt
        return self.loss_readback.read(st[:4], lambda s: (s[0] + lo * s[1] + ln * s[2]) / s[3])

    def save_models(self, train_step):
        num = str(train_step // self.args.save_cycle)
        if not os.path.exists(self.model_dir):
            os.makedirs(self.model_dir)
        cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
        self.eval_net.save_models(self.model_dir + '/' + num + '_rnn_net_params.pkl')
        torch.save(cpu(self.mixer), self.model_dir + '/' + num + '_mixer_net_params.pkl')
        torch.save(cpu(self.v), self.model_dir + '/' + num + '_v_net_params.pkl')

    def load_models(self):
        if os.path.exists(self.model_dir + '/rnn_net_params.pkl'):
            path_rnn = self.model_dir + '/rnn_net_params.pkl'
            path_mix = self.model_dir + '/mixer_net_params.pkl'
            path_v = self.model_dir + '/v_net_params.pkl'
            self.eval_net.load_models(path_rnn)
            self.mixer.load_state_dict(torch.load(path_mix, map_location='cpu'))
            self.v.load_state_dict(torch.load(path_v, map_location='cpu'))
            self.sync_replicas()
            print('Successfully load the model: {} and {}'.format(path_rnn, path_mix))
        else:
            raise Exception("No model!")

    def get_q_and_q_tot_table(self):
        """reference :237-272 - note the reference accumulates the one-hot across (i,j) iterations
        (u_onehot is modified in place and never reset); reproduced."""
        one = {'o': np.ones((1, 1, 2, 1)), 's': np.ones((1, 1, 1)), 'o_next': np.ones((1, 1, 2, 1)),
               'u_onehot': np.zeros((1, 1, 2, 3))}
        self.eval_net.init_hidden(1)
        q_values, _ = self.eval_net.get_current_q_values(one, 1)
        qv = q_values.cpu()
        q_table_i, q_table_j = qv[0, 0, 0].numpy(), qv[0, 0, 1].numpy()
        q_tot_table = np.zeros((3, 3))
        states = torch.ones(1, 1, 1)
        hidden = torch.zeros(1, 1, self.args.n_agents, self.args.rnn_hidden_dim)
        u_onehot = torch.zeros(1, 1, 2, 3)
        for i in range(3):
            for j in range(3):
                u_onehot[:, :, 0, i] = 1
                u_onehot[:, :, 1, j] = 1
                q_tot_table[i, j] = self._dense_joint_q(states, hidden, u_onehot)
        return q_tot_table, q_table_i, q_table_j

    def _dense_joint_q(self, states, hidden, u_dense):
        """joint Q for a (possibly multi-hot) dense action encoding, as the diagnostic above needs."""
        a = self.args
        dev = self.device
        from ..network.mixer import _linears
        from ..hostutil import lin_of, to_dev
        N, H, A, Q = a.n_agents, a.rnn_hidden_dim, a.n_actions, a.qtran_hidden_dim
        e0, e2 = _linears(self.mixer.hidden_action_encoding)
        q0, q2, q4 = _linears(self.mixer.q)
        ha = torch.cat([hidden, u_dense], dim=-1).reshape(N, H + A)
        x = to_dev(ha, dev)
        e1 = torch.empty(N, H + A, device=dev); e2b = torch.empty(N, H + A, device=dev)
        esum = torch.empty(1, H + A, device=dev)
        y1 = torch.empty(1, Q, device=dev); y2 = torch.empty(1, Q, device=dev); out = torch.empty(1, 1, device=dev)
        lin_of(e0).fwd(ops.src(x), e1, N, act=1)
        lin_of(e2).fwd(ops.src(e1), e2b, N)
        ops.agent_sum(e2b, esum, 1, N, H + A)
        lin_of(q0).fwd(ops.src(to_dev(states.reshape(1, -1), dev), esum), y1, 1, act=1)
        lin_of(q2).fwd(ops.src(y1), y2, 1, act=1)
        lin_of(q4).fwd(ops.src(y2), out, 1)
        return float(out.item())
