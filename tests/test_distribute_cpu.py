"""N>1 path on CPU: two gloo processes run the work distribution exactly as bench.py --config mixed64 does on N GPUs -- rank 0 owns
the transponder table and the configuration and BROADCASTS them, every rank derives the same MODCOD-grouped assignment, decodes its
share (the CPU oracle stands in for the device here), and the BBFRAMEs + per-frame statistics are GATHERED to the egress rank, which
puts them back into input order.  The reassembled bytes must equal what one process produces for the whole list."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (rate index, short): the "transponders" of the test are FEC-only units of different codes = different MODCOD groups
UNITS = [(3, 1), (6, 1), (3, 1), (5, 1), (6, 1), (3, 1), (5, 1), (6, 1), (3, 1)]
FRAMES = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _decode_unit(orc, u, rate, short, width):
    """FRAMES frames of unit u through the oracle FEC -> (row of `width` bytes, valid byte count, stats)"""
    row = np.zeros(width, np.uint8)
    pos, stats = 0, []
    for f in range(FRAMES):
        bb, bits = orc.encode_frame(rate, short, 1000 * u + f)
        llr = orc.bits_to_llr(bits, 6.5, np.random.default_rng(77 * u + f))
        out = np.zeros(bb.size, np.uint8)
        c = np.zeros(1, np.int32)
        tr = orc.lib().orc_fec_decode_frame(rate, short, llr, 16, 0, out, c)
        assert np.array_equal(out, bb)
        row[pos:pos + bb.size] = out
        pos += bb.size
        stats += [tr, int(c[0])]
    return row, pos, stats


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    import orc
    pkg = g.load_package()
    D = __import__('importlib').import_module(pkg.__name__ + '.distribute')
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    dd = D.Distributor(dist, 'cpu', egress=0)
    # rank 0 owns the table + configuration; the others start with nothing
    table = cfg = None
    if rank == 0:
        table = [dict(id=u, modcod=10 * r + s, rate=r, short=s, weight=float(orc.fec_params(r, s)['N'])) for u, (r, s) in enumerate(UNITS)]
        cfg = dict(max_trials=16, frames=FRAMES, width=max(orc.fec_params(r, s)['kbch'] // 8 for r, s in UNITS) * FRAMES)
    table = dd.broadcast_object(table)
    cfg = dd.broadcast_object(cfg)
    prbs = torch.arange(64, dtype=torch.uint8) if rank == 0 else torch.zeros(64, dtype=torch.uint8)     # a "table in device memory"
    dd.broadcast_tensor(prbs)
    assert prbs.tolist() == list(range(64))
    assign = D.assign_transponders(table, world)
    mine = assign[rank]
    rows, counts, stats = [], [], []
    for u in mine:
        row, n, st = _decode_unit(orc, u, table[u]['rate'], table[u]['short'], cfg['width'])
        rows.append(row); counts.append(n); stats.append(st)
    payload = torch.from_numpy(np.stack(rows)) if rows else torch.zeros((0, cfg['width']), dtype=torch.uint8)
    out, cnt = dd.gather_units(mine, payload, torch.tensor(counts, dtype=torch.int32), len(table))
    st_payload = torch.tensor(stats, dtype=torch.int32).view(torch.uint8).reshape(len(mine), -1) if mine else torch.zeros((0, 8 * FRAMES), dtype=torch.uint8)
    st_out, _ = dd.gather_units(mine, st_payload, torch.full((len(mine),), 8 * FRAMES, dtype=torch.int32), len(table))
    tmax = dd.max_over_ranks(len(mine))
    dd.barrier()
    # an argument error on ONE rank (rows of a different width) must surface on EVERY rank, before anybody enters the gathers
    try:
        dd.gather_units([rank], torch.zeros((1, 5 if rank == 1 else 4), dtype=torch.uint8), torch.tensor([1], dtype=torch.int32), 2)
        raised = False
    except ValueError:
        raised = True
    dd.barrier()
    q.put(('raised', rank, raised))
    if rank == 0:
        q.put((assign, out.numpy(), cnt.numpy(), st_out.numpy().view(np.int32), tmax))
    dist.destroy_process_group()


def test_two_ranks_broadcast_assign_gather_reassemble():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import orc
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = [q.get(timeout=180) for _ in range(world + 1)]
    [p.join(60) for p in ps]
    assert sorted(x[1:] for x in got if x[0] == 'raised') == [(0, True), (1, True)]      # the width mismatch was reported by both ranks
    assign, out, cnt, st, tmax = [x for x in got if x[0] != 'raised'][0]
    assert all(p.exitcode == 0 for p in ps)
    # every unit exactly once, balanced to within one unit (9 equal-weight units of three codes on two ranks: a code group is split
    # only because the balance demands it; test_assignment_helpers shows the exclusive case)
    assert sorted(sum(assign, [])) == list(range(len(UNITS)))
    assert abs(len(assign[0]) - len(assign[1])) <= 1, assign
    assert tmax == max(len(a) for a in assign)
    # byte-exact reassembly in input order: equals a single-process run over the whole list
    width = out.shape[1]
    for u, (r, s) in enumerate(UNITS):
        row, n, stats = _decode_unit(orc, u, r, s, width)
        assert cnt[u] == n and np.array_equal(out[u], row), u
        assert st[u].tolist() == stats, u


def test_assignment_helpers(pkg):
    import importlib
    D = importlib.import_module(pkg.__name__ + '.distribute')
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            parts = [D.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[k][1] == parts[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    groups = D.shard_by_weight([3, 1, 1, 1, 2, 2], 2)
    assert sorted(sum(groups, [])) == list(range(6))
    # BASELINE config 4: 64 transponders cycling over 8 MODCODs -> on 2/4/8 ranks every MODCOD sits on exactly one rank, loads balanced
    mods = [4, 6, 7, 11, 12, 13, 14, 15]
    table = [dict(id=t, modcod=mods[t % 8], weight=1.0 + 0.05 * (t % 8)) for t in range(64)]
    for world in (1, 2, 4, 8):
        a = D.assign_transponders(table, world)
        assert sorted(sum(a, [])) == list(range(64))
        owner = {}
        for r, idx in enumerate(a):
            for i in idx:
                owner.setdefault(table[i]['modcod'], set()).add(r)
        assert all(len(v) == 1 for v in owner.values()), (world, owner)
        loads = [sum(table[i]['weight'] for i in idx) for idx in a]
        assert max(loads) <= 1.25 * (sum(loads) / world), (world, loads)
    # the table bench.py --config mixed64 really builds (weights = LDPC edges x iterations + 40 x PLFRAME symbols of each MODCOD)
    import bench
    table = []
    for t in range(64):
        m = bench.MIXED_MODCODS[t % len(bench.MIXED_MODCODS)]
        info = pkg.modcod_info(m, False, False)
        table.append(dict(id=t, modcod=m, weight=float(info['ldpc_edges']) * bench.ITERS + 40.0 * info['plframe_symbols']))
    for world in (2, 4, 8):
        a = D.assign_transponders(table, world)
        assert sorted(sum(a, [])) == list(range(64))
        loads = [sum(table[i]['weight'] for i in idx) for idx in a]
        assert max(loads) <= 1.25 * (sum(loads) / world), (world, loads)
    # more ranks than MODCOD groups: groups are split, every rank gets work
    a = D.assign_transponders([dict(id=t, modcod=4, weight=1.0) for t in range(16)], 8)
    assert all(len(x) == 2 for x in a)
    # single-process distributor: gather = scatter into input order
    import torch
    dd = D.Distributor(None, 'cpu')
    out, cnt = dd.gather_units([2, 0], torch.tensor([[5, 6], [7, 8]], dtype=torch.uint8), torch.tensor([2, 1], dtype=torch.int32), 3)
    assert out.tolist() == [[7, 8], [0, 0], [5, 6]] and cnt.tolist() == [1, 0, 2]
    assert dd.broadcast_object({'a': 1}) == {'a': 1}


def test_fleet_plan_behind_the_c_abi_places_like_the_python_harness(pkg):
    """dvbs2gpu_fleet_plan (csrc/fleet.hip: what a C++ host with several GPUs gets) against distribute.assign_transponders on random tables: same placement,
    transponder by transponder -- MODCOD groups on one member where the balance allows, the contiguous cut otherwise, degenerate tables included"""
    import importlib
    import random
    D = importlib.import_module(pkg.__name__ + '.distribute')
    rnd = random.Random(5)
    cases = [([], 3), ([14], 4), ([4] * 7, 3), ([4, 6] * 8, 8)]
    for _ in range(300):
        n = rnd.randint(1, 80)
        pool = rnd.sample([4, 6, 7, 11, 12, 13, 14, 15, 18, 20, 23, 27], rnd.randint(1, 8))
        cases.append(([rnd.choice(pool) for _ in range(n)], rnd.randint(1, 9)))
    for mods, world in cases:
        heavy = rnd.random() < 0.3
        table = [dict(modcod=m, weight=(1000.0 * m + (rnd.choice([0.0, 17.5, 300.0]) if heavy else 0.0))) for m in mods]
        for tol in (0.25, 0.05, 1.0):
            want = D.assign_transponders(table, world, tol)
            got = pkg.fleet_plan([t['modcod'] for t in table], [t['weight'] for t in table], world, tol)
            back = [sorted(i for i, r in enumerate(got) if r == k) for k in range(max(world, 1))]
            assert back == [sorted(x) for x in want], (mods, world, tol)
