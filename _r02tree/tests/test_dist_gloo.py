"""world_size-2 (gloo, CPU) test of the data-parallel exchange step: each rank back-propagates the
UN-normalised loss numerator of its shard, GradReducer all-reduces [gradients | numerator | sum(mask)]
once, and dividing by the GLOBAL sum(mask) reproduces the single-process gradient and loss
(SURVEY 8e exactness rule).  The local gradient producer here is the CPU oracle (allowed in tests);
on the GPU the same GradReducer is fed by the HIP kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import seeded, learners

from golden_cases import case_states


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _numerator_grads(args, agent, mixer, batch, T):
    st = learners.LearnerState(args, agent, mixer)
    _, inter = learners.q_forward(st, batch, T=T)
    named = st.named_params()
    gs = torch.autograd.grad(inter["num"], [p for _, p in named], allow_unused=True)
    flat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, (_, p) in zip(gs, named)])
    return flat, float(inter["num"].detach()), float(inter["den"])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from marl_amd.algorithm.common import GradReducer, N_STATS
    case = ("x", "2s3z", "qmix", 6, 6, [6, 2, 3, 5, 4, 6], {})
    args, agent, mixer, _, _ = case_states(case)
    full = seeded.make_batch(args, 6, seed=100, lengths=[6, 2, 3, 4, 2, 3])   # rank 1's shard is shorter
    shard = {k: v[rank * 3:(rank + 1) * 3] for k, v in full.items()}
    red = GradReducer()
    assert red.enabled
    T = red.max_int(learners.max_episode_len(shard["terminated"], args.episode_limit), torch.device("cpu"))
    g, num, den = _numerator_grads(args, agent, mixer, shard, T)
    buf = torch.cat([g, torch.tensor([num, den] + [0.0] * (N_STATS - 2))])
    red.allreduce_(buf)
    # global max_episode_len: rank 0's episodes never terminate, rank 1's stop after 4 and 2 steps ->
    # both ranks must get 4 (never-terminating episodes are ignored, quirk Q2), not episode_limit
    from marl_amd.hostutil import DeviceBatch
    term = torch.zeros(2, 9, 1)
    if rank == 1:
        term[0, 3:] = 1.0
        term[1, 1:] = 1.0
    Tg = DeviceBatch.first_terminated_len(term, 9, reducer=red)
    none = DeviceBatch.first_terminated_len(torch.zeros(2, 9, 1), 9, reducer=red)     # nobody terminates anywhere
    # replicas start from rank 0's values (learners call this for parameters, targets and optimizer state)
    w = torch.full((7,), float(rank + 1))
    red.broadcast_(w, None)
    assert torch.equal(w, torch.ones(7))
    q.put((rank, (T, Tg, none), buf.numpy()))
    dist.destroy_process_group()


def test_sharded_numerators_reduce_to_the_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, T0, b0), (_, T1, b1) = sorted(res, key=lambda x: x[0])
    assert T0 == T1 == (6, 4, 9)
    np.testing.assert_array_equal(b0, b1)                  # every rank holds the same reduced buffer
    case = ("x", "2s3z", "qmix", 6, 6, None, {})
    args, agent, mixer, _, _ = case_states(case)
    full = seeded.make_batch(args, 6, seed=100, lengths=[6, 2, 3, 4, 2, 3])
    g, num, den = _numerator_grads(args, agent, mixer, full, None)
    n = g.numel()
    np.testing.assert_allclose(b0[n], num, rtol=1e-5)
    np.testing.assert_allclose(b0[n + 1], den)
    np.testing.assert_allclose(b0[:n] / b0[n + 1], g.numpy() / den, atol=2e-5, rtol=1e-3)
    # averaging per-shard losses would be wrong here: shards have different numbers of valid steps
    shard_losses = []
    for r in range(2):
        shard = {k: v[r * 3:(r + 1) * 3] for k, v in full.items()}
        _, sn, sd = _numerator_grads(args, agent, mixer, shard, 6)
        shard_losses.append(sn / sd)
    assert abs(np.mean(shard_losses) - num / den) > 1e-3 * abs(num / den)
