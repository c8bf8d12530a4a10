"""N>1 path on CPU: two gloo processes shard the transponder list exactly like bench.py does on N GPUs,
decode their share with the CPU oracle standing in for the device, and the gathered result covers every
unit exactly once (no data-path collective: only the final gather of results for checking)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    import orc
    pkg = g.load_package()
    sharding = __import__('importlib').import_module(pkg.__name__ + '.sharding')
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    a, b = sharding.shard_range(n_units, rank, world)
    # each unit = one short FEC frame; "decode" with the oracle, checksum the BBFRAME
    sums = torch.zeros(n_units, dtype=torch.int64)
    rng = np.random.default_rng(0)
    for u in range(a, b):
        bb, bits = orc.encode_frame(3, 1, 10 + u)
        llr = orc.bits_to_llr(bits, 6.0, np.random.default_rng(u))
        out = np.zeros(bb.size, np.uint8)
        c = np.zeros(1, np.int32)
        orc.lib().orc_fec_decode_frame(3, 1, llr, 16, 0, out, c)
        assert np.array_equal(out, bb)
        sums[u] = int(out.astype(np.int64).sum()) + 1
    dist.barrier()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)     # disjoint shards: the sum is the concatenation
    t = torch.tensor([float(b - a)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((sums.tolist(), float(t.item())))
    dist.destroy_process_group()


def test_two_rank_sharding_covers_every_unit_once():
    import torch.multiprocessing as mp
    n_units, world = 7, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    [p.start() for p in ps]
    sums, tmax = q.get(timeout=120)
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert all(s > 0 for s in sums) and len(sums) == n_units
    assert tmax == 4.0                                  # ceil(7/2): the max-over-ranks share


def test_shard_helpers(pkg):
    import importlib
    sh = importlib.import_module(pkg.__name__ + '.sharding')
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            parts = [sh.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[k][1] == parts[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    groups = sh.shard_by_weight([3, 1, 1, 1, 2, 2], 2)
    assert sorted(sum(groups, [])) == list(range(6))
    assert abs(sum([3, 1, 1, 1, 2, 2][i] for i in groups[0]) - 5) <= 1
