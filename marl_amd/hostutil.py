"""Host-side plumbing shared by the network / learner mirrors: flat parameter storage,
a dense-layer helper that maps an nn.Linear onto the HIP kernels, and the device batch."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops


def require_cuda(what="the MARL hot path"):
    if not torch.cuda.is_available():
        raise RuntimeError("%s runs only on the MI355X HIP kernels (no CPU fallback); "
                           "torch.cuda.is_available() is False" % what)
    return torch.device("cuda", torch.cuda.current_device())


class FlatParams:
    """One contiguous fp32 device buffer (and one gradient buffer) behind a list of nn.Parameters.

    The optimizer kernel, the gradient all-reduce and target-network sync then touch a single
    buffer each.  Parameters keep their names/shapes, so state_dict()/deepcopy work as in torch.
    """

    def __init__(self, params, device, with_grad=True):
        self.params = list(params)
        # every tensor starts on a 16-byte boundary (float4 operand loads); the padding floats are
        # zero and stay zero (zero gradient -> no optimizer movement)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        n = off
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device) if with_grad else None
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = self.flat[off:off + k].view(p.shape)
            if with_grad:
                p.grad = self.grad[off:off + k].view(p.shape)
        self.n = n

    def zero_grad(self):
        self.grad.zero_()


def flatten_module(module: nn.Module, device, with_grad=False):
    fp = FlatParams(list(module.parameters()), device, with_grad=with_grad)
    module._flat = fp
    return fp


class Lin:
    """An nn.Linear (weight (N,K), bias (N)) driven through marl_linear / marl_linear_wgrad."""

    def __init__(self, weight, bias, bf16=False):
        self.w, self.b, self.bf16 = weight, bias, bool(bf16)
        self.N, self.K = weight.shape

    def fwd(self, x, Y, M, act=0, beta=0.0):
        ops.linear(x, self.w.data, self.b.data if self.b is not None else None, Y, M, self.N, self.K, act=act, beta=beta,
                   bf16=self.bf16)

    def bwd_x(self, dY, dX, M, Yact=None, beta=0.0):
        """dX[M,K] (=|+=) (dY * relu'(Yact)) W"""
        ops.linear(ops.src(dY, gate=Yact), self.w.data, None, dX, M, self.K, self.N, beta=beta, w_kmajor=True,
                   bf16=self.bf16)

    def wgrad(self, dY, x, M, Yact=None):
        ops.linear_wgrad(dY, x, self.w.grad, self.b.grad if self.b is not None else None, M, self.N, self.K, Yact=Yact,
                         bf16=self.bf16)


def lin_of(module: nn.Linear, bf16=False):
    return Lin(module.weight, module.bias, bf16)


def to_dev(x, device, dtype=torch.float32):
    """numpy / torch (any device, any dtype) -> contiguous device tensor of dtype."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x)).to(device=device, dtype=dtype).contiguous()


class _PinnedRing:
    """Host -> device copies that do not stall the host: a pageable source makes the copy wait (on the host) until
    everything queued before it has run, which keeps the host from enqueuing the next kernels behind a long one.
    Small arrays go through a ring of pinned staging buffers instead; a slot is reused only after its copy finished."""
    SLOTS = 8

    def __init__(self):
        self.rings = {}

    def to_device(self, arr, device, dtype):
        a = np.ascontiguousarray(arr)
        t = torch.from_numpy(a)
        if t.dtype != dtype:
            t = t.to(dtype)
        key = (dtype, t.numel())
        ring = self.rings.setdefault(key, {"k": 0, "slots": []})
        if len(ring["slots"]) < self.SLOTS:
            ring["slots"].append([torch.empty(t.numel(), dtype=dtype).pin_memory(), None])
        slot = ring["slots"][ring["k"] % len(ring["slots"])] if len(ring["slots"]) == self.SLOTS else ring["slots"][-1]
        ring["k"] += 1
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0].copy_(t.reshape(-1))
        out = slot[0].to(device, non_blocking=True).reshape(t.shape)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return out


_PINNED = _PinnedRing()


def h2d_async(arr, device, dtype):
    """numpy array -> device tensor through pinned staging (see _PinnedRing); CPU targets just convert."""
    if torch.device(device).type != "cuda":
        return torch.as_tensor(np.asarray(arr), dtype=dtype, device=device)
    return _PINNED.to_device(arr, device, dtype)


class AsyncInt:
    """Read-back of a one-element device tensor that does not wait for what was enqueued AFTER its producer: the copy runs
    on a side stream behind an event recorded right after the producing kernel.  `wait()` blocks on that copy only."""
    _side = {}
    _bufs = {}

    def __init__(self, dev_tensor):
        cur = torch.cuda.current_stream()
        key = dev_tensor.device
        side = AsyncInt._side.get(key)
        if side is None:
            side = AsyncInt._side[key] = torch.cuda.Stream(device=key)
            AsyncInt._bufs[key] = [[torch.empty(1, dtype=dev_tensor.dtype).pin_memory() for _ in range(8)], 0]
        ring = AsyncInt._bufs[key]
        self.buf = ring[0][ring[1] % 8]
        ring[1] += 1
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            self.buf.copy_(dev_tensor, non_blocking=True)
            self.done = torch.cuda.Event()
            self.done.record(side)

    def wait(self):
        self.done.synchronize()
        return int(self.buf[0])


def onehot_to_index(u_onehot):
    """(…,A) one-hot or all-zero rows -> int32 index, -1 for all-zero rows (padding / t=0)."""
    s = u_onehot.sum(-1)
    idx = u_onehot.argmax(-1).to(torch.int32)
    return torch.where(s > 0, idx, torch.full_like(idx, -1)).contiguous()


class DeviceBatch:
    """Kernel-ready view of the 11-key episode dict (reference rollout.py:135-146).

    obs is addressed as (tensor, rows-per-episode, t0) so that (T+1)-slot storage serves both the
    current (t0=0) and next (t0=1) passes without materialising o_next.
    """

    def __init__(self):
        self.extra = {}
        self.ep_len = None      # per-episode lengths when obs is (T+1)-slot storage

    @staticmethod
    def first_terminated_len(term, episode_limit, reducer=None):
        """get_max_episode_len (algorithm/q_learner.py:49-66) on device: max over episodes of the first
        terminated index + 1; episodes that never terminate are ignored; 0 -> episode_limit.
        With a data-parallel ``reducer`` the max runs over the episodes of ALL ranks (shards must agree on
        T): the raw per-rank value (0 = none terminated) is all-reduced on the device before the single readback, so a
        rank whose episodes never terminate does not force episode_limit on the others."""
        dist_on = reducer is not None and reducer.enabled
        if term.is_cuda and term.dtype == torch.float32 and term.shape[0] > 0:
            out = ops.first_terminated_len(term, episode_limit)              # one kernel
            if dist_on:
                reducer.dist.all_reduce(out, op=reducer.dist.ReduceOp.MAX, group=reducer.group)
            m = int(out.item())                                              # one sync
            return m if m > 0 else episode_limit
        t = (term.reshape(term.shape[0], -1)[:, :episode_limit] == 1)
        anyt = t.any(dim=1)
        first = t.to(torch.int32).argmax(dim=1) + 1
        m = int(torch.where(anyt, first, torch.zeros_like(first)).max().item()) if t.shape[0] > 0 else 0
        if dist_on:
            m = reducer.max_int(m, torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                else torch.device("cpu"))
        return m if m > 0 else episode_limit

    @classmethod
    def from_dict(cls, batch, args, device, T=None):
        self = cls()
        N, O, S, A = args.n_agents, args.obs_shape, args.state_shape, args.n_actions
        term_full = to_dev(batch["terminated"], device)
        if T is None:
            T = cls.first_terminated_len(term_full, args.episode_limit)
        B = term_full.shape[0]
        self.B, self.T, self.N, self.O, self.S, self.A = B, T, N, O, S, A
        cut = lambda k, dt=torch.float32: to_dev(batch[k][:, :T], device, dt)
        o, o_next = cut("o"), cut("o_next")
        self.o_cur = (o, T * N, 0)
        self.o_next = (o_next, T * N, 0)
        def rows16(x):
            """(B*T, S) view whose rows start on 16-byte boundaries (row stride rounded up to 4 floats, zero pad): the
            fused mixer kernels take their vector-load paths for any S (MMM2: 322)"""
            x = x.view(B * T, S)
            if S % 4 == 0:
                return x
            buf = torch.zeros(B * T, (S + 3) // 4 * 4, dtype=x.dtype, device=x.device)
            buf[:, :S] = x
            return buf[:, :S]
        self.s = rows16(cut("s"))
        self.s_next = rows16(cut("s_next"))
        self.u_act = cut("u", torch.int32).view(B, T, N)
        if "u_idx" in batch:
            self.u_fed = cut("u_idx", torch.int32).view(B, T, N)
        else:
            self.u_fed = onehot_to_index(cut("u_onehot")).view(B, T, N)
        self.u_bs = T * N
        self.u_taken = self.u_fed
        self.r = cut("r").view(B * T)
        self.term = term_full[:, :T].contiguous().view(B * T)
        self.padded = cut("padded").view(B * T)
        self.avail = cut("avail_u").view(B * T * N, A)
        self.avail_next = cut("avail_u_next").view(B * T * N, A)
        return self

    @classmethod
    def from_record_auto(cls, rec, args, reducer=None, index=None, small=None):
        """from_record with max_episode_len found on the way: the kernel that computes it is launched first, the batch
        is prepared for the common case T = the record's full length while the GPU works, and only then the value is
        read back (the one host sync of an update) - the preparation's own launches are not left waiting behind it.
        A shorter T (every episode of the batch ended early) rebuilds the batch."""
        src = small if small is not None else rec
        term = src.term
        if not (term.is_cuda and term.dtype == torch.float32 and term.shape[0] > 0):
            T = cls.first_terminated_len(term, args.episode_limit, reducer=reducer)
            return cls.from_record(rec, args, T=T, index=index, small=small)
        db, pending = cls.from_record_begin(rec, args, reducer=reducer, index=index, small=small)
        T = pending()
        if T != db.T:
            db = cls.from_record(rec, args, T=T, index=index, small=small)
        return db

    @classmethod
    def from_record_begin(cls, rec, args, reducer=None, index=None, small=None):
        """First half of from_record_auto for device records: launches the max_episode_len kernel, starts its read-back on a
        side stream and builds the batch for the common case T = the record's full length.  Returns (db, pending);
        pending() is the agreed T (blocks only until the kernel and its copy are done, NOT on work enqueued since - a learner
        may launch its forward / backward on `db` first and redo it in the rare case T differs)."""
        src = small if small is not None else rec
        out = ops.first_terminated_len(src.term, args.episode_limit)
        if reducer is not None and reducer.enabled:
            reducer.dist.all_reduce(out, op=reducer.dist.ReduceOp.MAX, group=reducer.group)
        guess = min(rec.T, args.episode_limit)
        handle = AsyncInt(out)
        db = cls.from_record(rec, args, T=guess, index=index, small=small)

        def pending():
            m = handle.wait()
            return m if m > 0 else guess      # none terminated: episode_limit, never more steps than the record holds
        return db, pending

    @classmethod
    def from_record(cls, rec, args, T=None, index=None, small=None):
        """Zero-copy view of a device EpisodeRecord ((T+1)-slot storage): observations are read in
        place for both passes; only the small per-step arrays are re-packed when T < episode_limit.
        ``index`` (int64/int32 episode indices into ``rec``, e.g. a replay sample): the big arrays (obs, state)
        are read in place through an episode map, only the small arrays are gathered."""
        self = cls()
        Ta, N, O, S, A = rec.T, rec.N, rec.O, rec.S, rec.A
        big = rec
        self.o_map = None
        idx = None
        if index is not None:
            idx = index.to(device=rec.obs.device, dtype=torch.long)
            rec = small if small is not None else rec.select_small(idx)
            self.o_map = rec.o_map if getattr(rec, "o_map", None) is not None else idx.to(torch.int32).contiguous()
        E = rec.E
        if T is None:
            T = cls.first_terminated_len(rec.term, args.episode_limit)
        if T > Ta:
            raise ValueError("max_episode_len %d exceeds the %d steps the episode record holds "
                             "(args.episode_limit != record length?)" % (T, Ta))
        self.B, self.T, self.N, self.O, self.S, self.A = E, T, N, O, S, A
        self.o_cur = (big.obs, (Ta + 1) * N, 0)
        self.o_next = (big.obs, (Ta + 1) * N, 1)
        self.ep_len = rec.length
        st2 = big.state_store.view(big.E * (Ta + 1), -1)[:, :S]      # row stride padded to 16 bytes (EpisodeRecord)
        self.s = ops.Rows(st2, (T, Ta + 1, 0), self.o_map)
        self.s_next = ops.Rows(st2, (T, Ta + 1, 1), self.o_map)
        self.u_fed = rec.u
        self.u_bs = Ta * N
        cutc = lambda x: x[:, :T].contiguous()
        self.u_taken = cutc(rec.u)
        fused = getattr(rec, "avail_next", None) is not None       # select_small on the device (one launch) made these
        self.u_act = cutc(rec.u_act) if fused else self.u_taken.clamp(min=0)
        self.r, self.term, self.padded = cutc(rec.r).view(-1), cutc(rec.term).view(-1), cutc(rec.padded).view(-1)
        # `avail` (current-step availability, QPLEX / QTRAN only): gathered with the small arrays when the learner asked for
        # it (select_small(avail_cur=True)), else built lazily; a fused gather did not copy the full array
        if fused and getattr(rec, "avail_cur", None) is not None:
            self._avail = rec.avail_cur[:, :T].reshape(E * T * N, A)
        self._avail_src = (rec, T) if rec.avail is not None else (big, T, idx, rec.length)
        self.avail_next = (rec.avail_next[:, :T] if fused else rec.avail[:, 1:T + 1]).reshape(E * T * N, A)
        return self

    @property
    def avail(self):
        if "_avail" not in self.__dict__:
            if len(self._avail_src) == 2:
                rec, T = self._avail_src
                av, length = rec.avail[:, :T], rec.length
            else:
                big, T, idx, length = self._avail_src
                av = big.avail.index_select(0, idx)[:, :T]
            t_idx = torch.arange(T, device=av.device)[None, :, None, None]
            live = t_idx < length[:, None, None, None]
            self._avail = torch.where(live, av, torch.zeros((), device=av.device)).reshape(-1, av.shape[-1])
        return self._avail

    @avail.setter
    def avail(self, v):
        self._avail = v


def pin_to_gpu_numa(device_index=0):
    """Bind this process to the CPUs of the NUMA node the GPU hangs off (one process per GPU: kernel launches and the
    small D2H reads of an update are MMIO / PCIe round trips, several times slower from the other socket of a
    two-socket host).  Returns the node, or None when the topology cannot be read (then nothing is changed)."""
    import glob
    import os
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None
