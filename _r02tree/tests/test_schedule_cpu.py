"""Host-side launch schedule of the learner's three agent unrolls (no GPU needed)."""
from marl_amd.algorithm.common import PairedUnroll


def test_chain_split_model():
    p = PairedUnroll()
    T, O = 120, 80
    # 512 envs x 5 agents = 160 row tiles: the eval -> continuation chain gets one tile per workgroup (the pipelined
    # kernel) on 160 CUs, the target unroll two tiles per workgroup on the other 96
    assert p.chain_split(512 * 5, T, O) == (160, 96)
    a, b = p.chain_split(1024 * 5, T, O)
    assert a + b == 256 and a >= 128
    # large shards keep the plain schedule (the chain over few CUs would be longer than pair + continuation)
    assert p.chain_split(2048 * 5, T, O) is None
    assert p.chain_split(4096 * 5, T, O) is None
    # tiny batches and single steps are never split
    assert p.chain_split(16, T, O) is None and p.chain_split(512 * 5, 1, O) is None
    # wide observations cap the row tiles a workgroup can hold: no split that needs more
    assert p.chain_split(1024 * 10, T, 176) is None or max(-(-640 // c) for c in p.chain_split(1024 * 10, T, 176)) <= 2


def test_run_chain_falls_back_to_plain_order_without_continuation():
    calls = []
    p = PairedUnroll()
    p.enabled = False          # no GPU here: the disabled pair runs everything in order on the current stream
    p.run_chain(100, 8, 80, lambda cu: calls.append(("first", cu)), lambda cu: calls.append(("cont", cu)),
                lambda cu: calls.append(("second", cu)))
    assert calls == [("first", 256), ("second", 256), ("cont", 256)]
    calls.clear()
    p.run_chain(100, 8, 80, lambda cu: calls.append(("first", cu)), None, lambda cu: calls.append(("second", cu)))
    assert calls == [("first", 256), ("second", 256)]


def test_saved_activation_tile_layout_decode():
    """ops.saved_plane inverts the kernels' tile layout [T][tile][plane][column tile c][lane = 16 q + m][i]:
    element (row = 16 tile + 4 q + i, column = 16 c + m) sits at (((t * NT + tile) * P + plane) * 4 + c) * 256 + (16 q + m) * 4 + i
    (csrc/agent.hip: sv_off)."""
    import torch
    from marl_amd import ops
    T, B, N = 2, 3, 7                    # 21 rows -> two 16-row tiles
    shape = ops.saved_shape(T, B, N)
    assert shape == (T + 1, 32, 6, 64)   # one more slab: the hidden state after the last step
    assert ops.saved_shape(T, B, N, planes=3) == (T, 32, 3, 64)
    buf = torch.arange(int(torch.tensor(shape).prod()), dtype=torch.float32).reshape(shape)
    NT, P = 2, 6
    for plane in (0, 4):
        got = ops.saved_plane(buf, plane, B * N)
        assert got.shape == (T + 1, B * N, 64)
        for (t, row, col) in ((0, 0, 0), (1, 5, 17), (2, 20, 63), (1, 16, 32)):
            tile, q, i, c, m = row // 16, (row % 16) // 4, row % 4, col // 16, col % 16
            off = (((t * NT + tile) * P + plane) * 4 + c) * 256 + (16 * q + m) * 4 + i
            assert float(got[t, row, col]) == float(off)
