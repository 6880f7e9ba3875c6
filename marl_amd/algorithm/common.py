"""Shared machinery of the two learners: flat parameter/gradient storage with a statistics
tail, the fused clip+optimizer wrapper, the agent backward pass and data-parallel reduction."""
from __future__ import annotations

import warnings
import weakref

import torch

from .. import ops, experiments
from ..hostutil import FlatParams

MASK_BIG = -9999999.0        # reference algorithm/q_learner.py:105,112,126 ; qtran_learner.py:106
MASK_QTRAN_EVAL = -999999.0  # reference algorithm/qtran_learner.py:105
N_STATS = 4                  # tail of the gradient buffer: loss numerators + sum(mask)


class FlatView:
    """A slice of a FlatParams buffer that belongs to one module (for single-copy target sync)."""

    def __init__(self, flat, params, start):
        self.params = list(params)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4      # same 16-byte alignment rule as FlatParams
        self.n = off
        self.flat = flat[start:start + self.n]


class LearnerParams(FlatParams):
    """All trainable parameters of a learner in ONE buffer; the gradient buffer carries N_STATS
    extra floats so that gradients and loss statistics travel in a single all-reduce."""

    def __init__(self, params, device):
        params = list(params)
        super().__init__(params, device, with_grad=False)
        self.gradx = torch.zeros(self.n + N_STATS, dtype=torch.float32, device=device)
        self.grad = self.gradx[:self.n]
        self.stats = self.gradx[self.n:]
        for p, off in zip(self.params, self.offsets):
            p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def zero_grad(self):
        self.gradx.zero_()


class FusedOptimizer:
    """clip_grad_norm_ + RMSprop / Adam (torch defaults; reference q_learner.py:42-47,170-173) as
    two kernel launches over the flat buffer.  ``step(den)`` divides the gradient by den[0]
    (= global sum(mask)) inside the kernel."""

    def __init__(self, flat: LearnerParams, kind, lr, clip):
        if kind not in ("RMS", "Adam"):
            raise ValueError("optimizer {} not recognised.".format(kind))
        self.flat, self.kind, self.lr, self.clip = flat, kind, lr, clip
        dev = flat.flat.device
        self.s1 = torch.zeros(flat.n, device=dev)
        self.s2 = torch.zeros(flat.n, device=dev) if kind == "Adam" else None
        self.sumsq = torch.zeros(1, device=dev)
        self.t = 0
        self.param_groups = [{"params": flat.params, "lr": lr}]

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, den=None):
        f = self.flat
        self.t += 1
        ops.grad_sumsq(f.grad, f.n, self.sumsq)
        if self.kind == "RMS":
            ops.rmsprop_step(f.flat, f.grad, self.s1, f.n, self.lr, 0.99, 1e-8, self.clip, self.sumsq, den)
        else:
            ops.adam_step(f.flat, f.grad, self.s1, self.s2, f.n, self.lr, 0.9, 0.999, 1e-8,
                          1.0 - 0.9 ** self.t, (1.0 - 0.999 ** self.t) ** 0.5, self.clip, self.sumsq, den)

    def state_dict(self):
        return {"kind": self.kind, "t": self.t, "s1": self.s1.cpu(), "s2": None if self.s2 is None else self.s2.cpu()}

    def load_state_dict(self, sd):
        self.t = sd["t"]
        self.s1.copy_(sd["s1"])
        if self.s2 is not None and sd["s2"] is not None:
            self.s2.copy_(sd["s2"])


class ResumeMixin:
    """Full-resume state the reference's checkpoints lack: parameters, BOTH target networks, the
    optimizer's running statistics and step count.  ``save_models`` / ``load_models`` keep the reference's three
    state-dict files (q_learner.py:193-209); these two methods add one extra file next to them."""

    def _target_flats(self):
        out = {"target_agent": self.target_net.agent._flat.flat}
        if self.target_mixer is not None and getattr(self.target_mixer, "_flat", None) is not None and self.target_mixer._flat.n:
            out["target_mixer"] = self.target_mixer._flat.flat
        return out

    def resume_state(self):
        sd = {"alg": self.args.alg, "n_params": int(self._flat.n), "params": self._flat.flat.detach().cpu().clone(),
              "optimizer": self.optimizer.state_dict()}
        for k, t in self._target_flats().items():
            sd[k] = t.detach().cpu().clone()
        return sd

    def load_resume_state(self, sd):
        if sd["alg"] != self.args.alg or sd["n_params"] != int(self._flat.n):
            raise ValueError("resume state of a different learner (%s, %d parameters)" % (sd["alg"], sd["n_params"]))
        self._flat.flat.copy_(sd["params"])
        for k, t in self._target_flats().items():
            t.copy_(sd[k])
        self.optimizer.load_state_dict(sd["optimizer"])
        self.sync_replicas()

    def save_resume(self, path):
        torch.save(self.resume_state(), path)

    def load_resume(self, path):
        self.load_resume_state(torch.load(path, map_location="cpu"))


class LossReadback:
    """The loss of an update as a host float.  Default: read now (one blocking copy, as the reference's `loss.item()`
    use).  With ``args.lazy_loss = True`` train() returns a handle instead: the statistics are copied to pinned memory
    in stream order and `float(handle)` waits only for that copy - the host goes on to enqueue the next rollout / update
    while this one still runs, which is worth ~0.2 ms per step on small shards (nothing idles between the updates).
    The arithmetic is the same either way (fp32 on the host)."""

    class Handle:
        __slots__ = ("buf", "event", "fn", "_v")

        def __init__(self, buf, event, fn):
            self.buf, self.event, self.fn, self._v = buf, event, fn, None

        def __float__(self):
            if self._v is None:
                self.event.synchronize()
                self._v = float(self.fn(self.buf))
            return self._v

    RING = 8      # an un-read handle stays valid for this many later updates

    def __init__(self, args):
        self.lazy = bool(getattr(args, "lazy_loss", False))
        self.slots, self.k = [], 0

    def read(self, stats, fn):
        """stats: device vector; fn: host tensor -> loss value."""
        if not self.lazy:
            return float(fn(stats.cpu()))        # one copy + sync; the division runs on the host in fp32
        if len(self.slots) < self.RING:
            self.slots.append(torch.empty(stats.numel(), dtype=stats.dtype).pin_memory())
        buf = self.slots[self.k % self.RING]
        self.k += 1
        buf.copy_(stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return LossReadback.Handle(buf, ev, fn)


class SpeculativeBatchMixin:
    """Learners with `_forward_backward(db)`, `args`, `reducer`, `max_episode_len`."""
    _full_len_streak = 0      # consecutive updates whose max_episode_len was the record's full length

    def _device_batch(self, rec, index, small):
        """DeviceBatch of a device record with max_episode_len agreed.  While the last updates all ran at the record's
        full length, the forward / backward is launched for that length BEFORE the value is read back (the read-back waits
        only for the kernel that computes it): the host never waits for the device in front of an update, and the device
        never waits for the host's first launches after the sync.  A different length - every episode of the batch ended
        early - redoes the pass (forward / backward overwrite their outputs and zero the gradient buffer themselves).
        Returns None when the pass has already been launched."""
        from ..hostutil import DeviceBatch
        term = (small if small is not None else rec).term
        if not (term.is_cuda and term.dtype == torch.float32 and term.shape[0] > 0) or self._full_len_streak < 2:
            db = DeviceBatch.from_record_auto(rec, self.args, reducer=self.reducer, index=index, small=small)
            full = db.T == min(rec.T, self.args.episode_limit)
            self._full_len_streak = self._full_len_streak + 1 if full else 0
            return db
        db, pending = DeviceBatch.from_record_begin(rec, self.args, reducer=self.reducer, index=index, small=small)
        self.max_episode_len = db.T
        self._forward_backward(db)
        T = pending()
        if T == db.T:
            return None
        self._full_len_streak = 0
        return DeviceBatch.from_record(rec, self.args, T=T, index=index, small=small)


class Scratch:
    def __init__(self):
        self.d = {}

    def get(self, name, shape, device, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self.d.get(key)
        if t is None or t.device != device:
            t = torch.empty(*shape, dtype=dtype, device=device)
            self.d[key] = t
        return t


def agent_backward(mac, db, which, saved, hs, dq, dhs, buf, dq_idx=None, dq_val=None, dq_idx2=None, dq_val2=None, dq_gdiv=1):
    """BPTT of the eval unroll: the fused kernel (delta pass + W_ih/W_hh/W_2 gradients), then the
    fc1 weight gradient as one reduction over the virtual input [obs | one-hot(u_{t-1}) | agent id]
    (autograd of controller/share_params.py:125-146 + network/q_network.py:16-21)."""
    args = mac.args
    B, T, N, A, O = db.B, db.T, db.N, db.A, db.O
    H = args.rnn_hidden_dim
    M = B * T * N
    dev = saved.device
    dxp = buf.get("dxp", (B, T, N, H), dev)
    w = mac.agent.weights()
    ag = mac.agent
    grads = {"rnn.weight_ih": ag.rnn.weight_ih.grad, "rnn.weight_hh": ag.rnn.weight_hh.grad,
             "rnn.bias_ih": ag.rnn.bias_ih.grad, "rnn.bias_hh": ag.rnn.bias_hh.grad,
             "fc2.weight": ag.fc2.weight.grad, "fc2.bias": ag.fc2.bias.grad}
    # opt-in args.gemm_mode = "bf16x6": the split BPTT kernel (csrc/agent_bwd_x6.hip; one workgroup per 16 rows up to 256 row tiles,
    # per 32 rows beyond) - faster than the fp32 kernels at every size measured (profiles/archive/r04_unroll_x6_times.txt)
    from ..network import mixer as _mixer
    x6 = (getattr(args, "gemm_mode", _mixer.DEFAULT_GEMM_MODE) == "bf16x6" and dq is None and dq_idx is not None
          and (B * N + 31) // 32 >= experiments.get("x6_bwd_min_wg") and ops.agent_unroll_bwd_x6_supported(B, T, N, A))
    ops.agent_unroll_bwd(w, dq, dhs, saved, hs, dxp, None, grads, B, T, N, A, dq_idx=dq_idx, dq_val=dq_val,
                         dq_idx2=dq_idx2, dq_val2=dq_val2, dq_gdiv=dq_gdiv, x6=x6)
    obs, obs_bs, obs_t0 = db.o_cur if which == "cur" else db.o_next
    remap0 = None if (obs_bs == T * N and obs_t0 == 0) else (T * N, obs_bs, obs_t0 * N)
    kw = {}
    if args.last_action:
        kw.update(idx=db.u_fed.reshape(-1, 1), nhot=1, hot_w=A, remapi=(T * N, db.u_bs, (-1 if which == "cur" else 0) * N))
    emap = getattr(db, 'o_map', None)
    if emap is not None and remap0 is None:
        remap0 = (T * N, obs_bs, obs_t0 * N)
    xin = ops.src(obs.reshape(-1, O), nid=N if args.reuse_network else 0, remap0=remap0, emap0=emap, **kw)
    I = O + (A if args.last_action else 0) + (N if args.reuse_network else 0)
    ops.linear_wgrad(dxp.view(M, H), xin, ag.fc1.weight.grad, ag.fc1.bias.grad, M, H, I, bf16=False)   # agent layers stay fp32


class GradReducer:
    """Data-parallel exchange step: ONE all-reduce(sum) of [gradients | loss numerators | sum(mask)]
    over RCCL/xGMI (SURVEY 8e exactness rule: un-normalised numerators are summed, the division by
    the GLOBAL sum(mask) happens afterwards in the optimizer kernel)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        up = dist.is_available() and dist.is_initialized()
        # experiments.force_reducer (MARL_FORCE_REDUCER=1): take the collective path with a single rank too (RCCL smoke test on a 1-GPU box)
        self.enabled = up and (dist.get_world_size(group) > 1 or experiments.get("force_reducer") == 1)

    def allreduce_(self, flat_with_stats):
        if self.enabled:
            self.dist.all_reduce(flat_with_stats, op=self.dist.ReduceOp.SUM, group=self.group)
        return flat_with_stats

    def broadcast_(self, *tensors, src=0):
        """replicas start from rank `src`'s values (parameters, targets, optimizer state): data-parallel training is
        only exact when every rank holds the same weights - never rely on identical seeding"""
        if self.enabled:
            for t in tensors:
                if t is not None:
                    self.dist.broadcast(t, src=src, group=self.group)

    def max_int(self, value, device):
        """global max of a host integer (used for the global max_episode_len)."""
        if not self.enabled:
            return value
        t = torch.tensor([value], dtype=torch.int64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())


class PairedUnroll:
    """Launches two independent agent unrolls (eval current-Q and target next-Q: reference q_learner.py:97,104) side by
    side: each is limited to half of the CUs (the cu_budget argument of marl_agent_unroll_fwd - per call, no process
    state) and the second runs on a side HIP stream.  A
    T-step unroll is a chain of T dependent steps whose latency barely depends on how many row tiles a workgroup
    carries (0.39 / 0.62 / 0.95 ms at 1 / 2 / 3 tiles), so below ~3000 episodes per GPU - the shards of the multi-GPU
    runs - two half-chip launches finish sooner than two whole-chip launches back to back.  Results do not depend on
    the split (rows are independent)."""

    MAX_TILES = 1024          # 128 workgroups x 8 row tiles of 16 rows (the LDS cap of the unroll kernel)

    def __init__(self, x6=False):
        self.side = None
        self.x6 = bool(x6)                                          # the unrolls run on agent_fwd_x6_kernel (one or two row tiles per workgroup)
        self.enabled = experiments.get("no_pair") != 1              # experiments: MARL_NO_PAIR=1 launches them back to back
        self.chain = experiments.get("no_chain") != 1               # experiments: MARL_NO_CHAIN=1 keeps the plain pair + continuation
        # experiments only: MARL_CHAIN_SPLIT=<CUs of the chain side> (marl_amd/experiments.py validates it), 0 = never chain
        self.forced_split = experiments.get("chain_split")
        self.big_pair = experiments.get("big_pair") == 1            # experiments: MARL_BIG_PAIR=1 (see run_chain)

    def applies(self, rows, T):
        return self.enabled and T >= 8 and 32 <= (rows + 15) // 16 <= self.MAX_TILES

    def run(self, rows, T, first, second):
        """first(cu), second(cu): closures that launch one unroll each on the current stream over `cu` CUs."""
        if not self.applies(rows, T):
            first(256)
            second(256)
            return
        self._fork(lambda: first(128), lambda: second(128))

    def _fork(self, main, side):
        cur = torch.cuda.current_stream()
        if self.side is None or self.side.device != cur.device:
            self.side = torch.cuda.Stream(device=cur.device)
        self.side.wait_stream(cur)              # inputs written on the main stream are visible to the side launch
        with torch.cuda.stream(self.side):
            side()
        main()
        cur.wait_stream(self.side)

    # measured step time of the unroll kernel by row tiles per workgroup (us per step, 2s3z-sized agent): one tile runs the
    # software-pipelined kernel; beyond that ~1.3 + 2.1 per tile (5.5 at 2, 7.6 at 3, 11.7 at 5)
    def _step_us(self, rt):
        if self.x6:
            # agent_fwd_x6_kernel (profiles/archive/r04_unroll_x6_times.txt: 0.18 / 0.33 / 1.04 ms per 120 steps at 1 / 2 / 5 row tiles per CU;
            # it holds one or two tiles per workgroup and runs more in rounds of workgroups)
            return 1.5 if rt <= 1 else 2.75 if rt <= 2 else 1.74 * rt
        return 3.3 if rt <= 1 else 1.3 + 2.1 * rt

    def chain_split(self, rows, T, obs_dim):
        """CU split (chain, side) for run_chain, or None when the plain schedule (pair, then the continuation over the
        whole chip) is at least as fast by the step-time model."""
        if not self.chain or not self.applies(rows, T):
            return None
        if self.forced_split is not None:                           # experiments: CUs of the chain side (the model's choice otherwise)
            return (self.forced_split, 256 - self.forced_split) if self.forced_split > 0 else None
        tiles = (rows + 15) // 16
        # row tiles per workgroup the fp32 unroll kernel can hold (the split kernel runs further tiles in rounds of workgroups)
        cap = 8 if self.x6 else max(1, min(8, 2048 // (4 * max(obs_dim, 4))))
        ceil = lambda a, b: -(-a // b)
        # (the continuation reads the input-side work the first unroll stored: ~0.6 of a full unroll's step time)
        plain = self._step_us(ceil(tiles, 128)) + 0.6 * self._step_us(ceil(tiles, 256))
        best = None
        for cu_a in (128, 144, 160, 176, 192):
            cu_b = 256 - cu_a
            rt_a, rt_b = ceil(tiles, cu_a), ceil(tiles, cu_b)
            if rt_a > cap or rt_b > cap:
                continue
            cost = max(1.6 * self._step_us(rt_a), self._step_us(rt_b))
            if best is None or cost < best[0]:
                best = (cost, cu_a, cu_b)
        if best is None or best[0] > 0.9 * plain:
            return None
        return best[1], best[2]

    def run_chain(self, rows, T, obs_dim, first, cont, second):
        """first -> cont is a dependent chain of two unrolls (eval current-Q, then its continuation over the next
        observations: quirk Q1), second is independent of both (target next-Q).  On small shards the chain runs on one
        stream over most of the CUs - few row tiles per workgroup, so the short-step kernel - while `second` runs beside
        it on the rest with more tiles per workgroup; nothing waits for a launch gap in the middle.  Larger shards keep
        the plain schedule (pair first/second, then cont over the whole chip)."""
        split = self.chain_split(rows, T, obs_dim) if cont is not None else None
        if split is None and self.big_pair and self.enabled and not self.applies(rows, T) and T >= 8:
            # batches beyond the pair's tile cap: the HBM-bound saving unroll and the issue-bound target unroll in flight
            # together, every launch over the whole chip - the dispatcher mixes their workgroups over the CUs
            if cont is not None:
                self._fork(lambda: (first(256), cont(256)), lambda: second(256))
            else:
                self._fork(lambda: first(256), lambda: second(256))
            return
        if split is None:
            self.run(rows, T, first, second)
            if cont is not None:
                cont(256)
            return
        cu_a, cu_b = split
        self._fork(lambda: (first(cu_a), cont(cu_a)), lambda: second(cu_b))


class GraphedUpdate:
    """hipGraph replay of a learner's forward/backward schedule for replay-ring samples of a fixed shape.  The ~25 kernel
    launches of ``_forward_backward`` become ONE graph launch; what varies between updates (the sampled episode indices and
    the small per-step arrays gathered from the ring) lives in persistent buffers the captured kernels point at.  Pays on
    small per-GPU shards - the shape of every rank of a multi-GPU run - where an update is a millisecond or two and the
    host-side launch path is what bounds it; the gradient all-reduce and the optimizer stay outside the graph.
    ``args.hip_graph``: True = always, False = never, absent / None / "auto" = for batches of at most AUTO_MAX_EPISODES
    episodes.  Falls back to eager launches whenever the shape differs, max_episode_len is shorter than the record, or
    capture is not possible.  Once the last updates all ran at the record's full length the graph is replayed BEFORE
    max_episode_len is read back (the read-back waits only for the kernel that computes it); a shorter length redoes the
    pass eagerly (forward / backward overwrite their outputs and zero the gradient buffer themselves)."""

    WARMUP = 2          # eager updates on the static buffers before capture (allocations, workspace growth)
    AUTO_MAX_EPISODES = 1536

    def __init__(self, auto=False):
        self.entries = {}
        self.disabled = False
        self.auto = bool(auto)
        self.replays = 0

    @staticmethod
    def from_args(args):
        """the learner's GraphedUpdate (or None) for args.hip_graph"""
        mode = getattr(args, "hip_graph", None)
        if mode is None or mode == "auto":
            return GraphedUpdate(auto=True)
        return GraphedUpdate() if mode else None

    SCHEDULE_ARGS = ("gemm_mode", "mixer_dtype", "mixer_wgrad_dtype", "double_q", "no_loss_fold", "lazy_loss", "gamma",
                     "two_hyper_layers", "last_action", "reuse_network")

    @staticmethod
    def _schedule_key(args):
        """what a captured schedule froze besides shapes and pointers: the generation of the experiments table (every
        experiments.set() bumps it) and the args fields the launch sequence reads"""
        from .. import experiments
        return (experiments.generation,) + tuple(repr(getattr(args, k, None)) for k in GraphedUpdate.SCHEDULE_ARGS)

    def run(self, learner, ring, index):
        """Returns True when the update's forward/backward was done here (static buffers + graph), else False."""
        from ..hostutil import DeviceBatch, AsyncInt
        self.prepared = None
        if self.disabled:
            return False
        args = learner.args
        if self.auto and int(index.numel()) > self.AUTO_MAX_EPISODES:
            return False
        # static buffers alias the ring's (E, T) arrays: only full-length records of the steady-state batch size
        if ring.T != args.episode_limit or int(index.numel()) != int(args.batch_size):
            return False
        key = (id(ring), int(index.numel()), ring.T)
        e = self.entries.get(key)
        if e is not None and e["ring"]() is not ring:      # another record reuses the id of a freed one
            e = None
        dev = ring.obs.device
        if e is None:
            # ONE persistent index tensor: the batch keeps a reference to it (the lazy `avail` gather of QPLEX reads the
            # ring through it), so refreshing it below is what every later warm-up, capture and replay sees
            idx = index.to(device=dev, dtype=torch.long).clone()
            small = ring.select_small(idx, avail_cur=getattr(learner, "needs_avail", False))
            db = DeviceBatch.from_record(ring, args, T=min(ring.T, args.episode_limit), index=idx, small=small)
            # a device gather (one launch) wrote u_act / avail_next / avail_cur itself and the batch holds full-length VIEWS of
            # them (T = ring.T here): refreshing `small` in place below refreshes the batch.  Other records get static copies.
            fused = getattr(small, "avail_next", None) is not None and db.T == ring.T
            e = self.entries[key] = dict(ring=weakref.ref(ring), idx=idx, small=small, db=db, calls=0, graph=None, streak=0,
                                         fused=fused, avail_next=None if fused else db.avail_next.clone(),
                                         u_act=None if fused else db.u_act.clone())
            if not fused:
                db.avail_next, db.u_act = e["avail_next"], e["u_act"]
            e["T"] = db.T
        idx, small, db = e["idx"], e["small"], e["db"]
        idx.copy_(index)
        db.o_map.copy_(idx)
        ring.select_small(idx, out=small)
        if e["graph"] is not None and (e["ws_gen"] != ops.WS.gen or e["sched_key"] != self._schedule_key(args)):
            # a workspace the captured kernels point at was reallocated (another learner / a larger request): the graph would
            # write into retired storage; or a host-side decision frozen at capture (an experiments switch, an args field the
            # schedule reads) changed on the live learner - drop it and capture again after a warm-up
            e["graph"], e["ws_keep"], e["calls"] = None, None, 0
        pending = None
        term = small.term
        if e["graph"] is not None and e["streak"] >= 2 and term.is_cuda and term.dtype == torch.float32 and term.shape[0] > 0:
            # speculative: launch the length kernel, start its read-back on a side stream, replay for the full length meanwhile
            out = ops.first_terminated_len(term, args.episode_limit)
            r = learner.reducer
            if r is not None and r.enabled:
                r.dist.all_reduce(out, op=r.dist.ReduceOp.MAX, group=r.group)
            pending = AsyncInt(out)
            T = e["T"]
        else:
            T = DeviceBatch.first_terminated_len(term, args.episode_limit, reducer=learner.reducer)
            if T != e["T"]:
                e["streak"] = 0
                self.prepared = (small, T)       # the eager path reuses the gathered arrays and the agreed T
                return False
            e["streak"] += 1
        if not e["fused"]:
            torch.clamp(small.u[:, :T], min=0, out=e["u_act"])
            e["avail_next"].view(small.E, T, small.N, small.A).copy_(small.avail_next[:, :T] if small.avail is None else small.avail[:, 1:T + 1])
        if getattr(small, "avail_cur", None) is None:
            db.__dict__.pop("_avail", None)
        learner.max_episode_len = T
        e["calls"] += 1
        if e["graph"] is None and e["calls"] > self.WARMUP:
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: API calls of other threads (the NCCL watchdog polling its events) do not break the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    learner._forward_backward(db)
                e["graph"] = g
                e["ws_gen"], e["ws_keep"] = ops.WS.gen, ops.WS.snapshot()
                e["sched_key"] = self._schedule_key(args)
            except Exception as ex:      # capture not possible on this stack: stay eager for good
                self.disabled = True
                self.error = repr(ex)
                warnings.warn("hipGraph capture of the learner update failed (%s): updates stay on eager launches" % self.error)
                torch.cuda.synchronize()
                learner._forward_backward(db)
                return True
        if e["graph"] is not None:
            e["graph"].replay()
            self.replays += 1
        else:
            learner._forward_backward(db)
        if pending is not None:
            m = pending.wait()
            Tr = m if m > 0 else e["T"]
            if Tr != e["T"]:                     # every episode of the batch ended early: redo at its own length, eagerly
                e["streak"] = 0
                self.prepared = (small, Tr)
                return False
        return True
