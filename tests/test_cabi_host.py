"""CPU-side checks of the shipped library: every symbol of include/dvbs2gpu.h is exported, the static
queries work without a GPU, the engine refuses to run without one (no CPU fallback), and the LDPC plan's
intra-layer ordering is consistent with the reference's sequential row order."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'dvbs2gpu.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dvbs2gpu_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    # and the ctypes prototype table covers the header
    assert set(names) <= set(pkg.PROTOTYPES)


def test_static_queries_without_gpu(pkg):
    mi = pkg.modcod_info(14, False, False)
    assert (mi['plframe_symbols'], mi['ldpc_n'], mi['ldpc_k'], mi['kbch'], mi['ldpc_edges'], mi['bch_t']) == (21690, 64800, 48600, 48408, 226799, 12)
    assert pkg.modcod_info(4)['plframe_symbols'] == 32490
    assert pkg.modcod_info(27, True, True)['plframe_symbols'] == 3402       # 36 slots + 2 pilot blocks
    for bad in (0, -1, 29, 100):
        with pytest.raises(pkg.Dvbs2GpuError) as e:
            pkg.modcod_info(bad)
        assert e.value.code == -2                                              # reference throws here (modcod_to_cfg.cpp:11,135)
    with pytest.raises(pkg.Dvbs2GpuError):
        pkg.modcod_info(11, True)                                              # short 9/10 has no table
    for modcod in range(1, 29):
        for short in (0, 1):
            o = orc.modcod_params(modcod, short, 1)
            if o is None:
                continue
            g = pkg.modcod_info(modcod, bool(short), True)
            assert (g['constellation'], g['bits_per_symbol'], g['rate'], g['slots'], g['pilot_blocks'], g['plframe_symbols'], g['ldpc_n'], g['kbch']) == \
                   (o['constel'], o['bits'], o['rate'], o['slots'], o['pilot_blocks'], o['plframe'], o['N'], o['kbch'])


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = C.c_void_p()
    rc = pkg.load_library().dvbs2gpu_create(0, C.byref(h))
    assert rc == -4 and not h.value
    with pytest.raises(pkg.Dvbs2GpuError):
        pkg.Engine(0)


@pytest.mark.parametrize('rate,short', orc.ALL_CODES)
def test_ldpc_plan_matches_reference_row_order(pkg, rate, short):
    lib = pkg.load_library()
    cnt = (C.c_int32 * 3)()
    assert lib.dvbs2gpu_ldpc_plan_dump(rate, short, None, None, None, cnt) == 0
    nl, ne, nr = list(cnt)
    layers = np.zeros((nl, 4), np.uint32); ents = np.zeros(ne, np.uint32); rows = np.zeros(max(nr, 1), np.uint32)
    assert lib.dvbs2gpu_ldpc_plan_dump(rate, short, layers.ctypes.data, ents.ctypes.data, rows.ctypes.data, cnt) == 0
    p = orc.fec_params(rate, short)
    R = p['N'] - p['K']
    q = R // 360
    assert nl == q
    pos = np.zeros((R, 64), np.uint16)
    cn = np.zeros(R, np.uint8)
    lib_o = orc.lib()
    lib_o.orc_ldpc_rows.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cnl = lib_o.orc_ldpc_rows(rate, short, None, None)
    pos = np.zeros((R, cnl), np.uint16)
    lib_o.orc_ldpc_rows(rate, short, pos.ctypes.data, cn.ctypes.data)
    j = np.arange(360)
    for i in range(q):
        off, deg, dn, row_off = (int(x) for x in layers[i])
        chain_d, deg = deg >> 16, deg & 0xffff
        depth, nc = dn & 0xffff, dn >> 16
        e = ents[off:off + deg]
        sp, r = (e & 0xffff).astype(np.int64), (e >> 16).astype(np.int64)
        bits = 360 * r[None, :] + (j[:, None] + sp[None, :]) % 360             # [row j][link k]
        # 1. same set of information bits per row as the reference's expansion
        want = pos[360 * i:360 * i + 360, :cn[360 * i]].astype(np.int64)
        assert deg == cn[360 * i]
        assert np.array_equal(np.sort(bits, axis=1), np.sort(want, axis=1))
        # 2. ordering: any two rows of the layer that share a bit are ordered by level, flags agree
        owner = {}
        conflict_links = set()
        late = np.zeros((360, deg), bool); early = np.zeros((360, deg), bool)
        for jj in range(360):
            for k in range(deg):
                owner.setdefault(int(bits[jj, k]), []).append((jj, k))
        for b, lst in owner.items():
            if len(lst) > 1:
                lst.sort()
                for a in range(len(lst)):
                    conflict_links.add(lst[a][1])
                    if a > 0:
                        late[lst[a]] = True
                    if a + 1 < len(lst):
                        early[lst[a]] = True
        if not conflict_links:
            assert depth == 1 and nc == 0
            continue
        assert conflict_links == set(range(nc)) and nc <= 12
        rw = rows[row_off:row_off + 360]
        lvl = (rw & 0xff).astype(np.int64)
        for k in range(nc):
            assert np.array_equal(((rw >> (8 + k)) & 1).astype(bool), late[:, k])
            assert np.array_equal(((rw >> (20 + k)) & 1).astype(bool), early[:, k])
        assert lvl.max() == depth and lvl.min() == 1
        for b, lst in owner.items():
            for a in range(1, len(lst)):
                assert lvl[lst[a][0]] > lvl[lst[a - 1][0]]        # the later row waits for the earlier one


@pytest.mark.parametrize('rate,short', orc.ALL_CODES)
def test_ldpc_plan_walk_lists(pkg, rate, short):
    """quad-walk layers (csrc/ldpc_plan.h): the step list behind the layer's row words names every row of level >= 2 exactly once, never more
    than 64 / (lanes per row) rows per step, all rows of a step on one level, levels in ascending order -- the single walker wave relies on exactly that"""
    lib = pkg.load_library()
    cnt = (C.c_int32 * 3)()
    assert lib.dvbs2gpu_ldpc_plan_dump(rate, short, None, None, None, cnt) == 0
    nl, ne, nr = list(cnt)
    layers = np.zeros((nl, 4), np.uint32); ents = np.zeros(ne, np.uint32); rows = np.zeros(max(nr, 1), np.uint32)
    assert lib.dvbs2gpu_ldpc_plan_dump(rate, short, layers.ctypes.data, ents.ctypes.data, rows.ctypes.data, cnt) == 0
    WALK = 0xfffe
    for i in range(nl):
        off, deg, dn, row_off = (int(x) for x in layers[i])
        if (deg >> 16) != WALK:
            continue
        depth, nc = dn & 0xffff, dn >> 16
        lvl = (rows[row_off:row_off + 360] & 0xff).astype(np.int64)
        nsteps, lpr = int(rows[row_off + 360]) & 0xffff, int(rows[row_off + 360]) >> 16      # lanes per row: 4, or 8 in the kernels for degree > 12
        assert lpr in (4, 8) and 1 <= nc <= lpr and depth >= 2
        lst = rows[row_off + 361:row_off + 361 + 16 * (nsteps + 3)].astype(np.int64).reshape(nsteps + 3, 16)
        assert np.all(lst[nsteps:] == 0xffffffff)                   # the empty steps the walker's look-ahead may fetch
        seen, last_level = [], 1
        for st in range(nsteps):
            r = lst[st][lst[st] != 0xffffffff]
            assert 1 <= r.size <= 64 // lpr and np.all(r < 360)
            lv = set(lvl[r].tolist())
            assert len(lv) == 1
            assert lv.pop() >= last_level
            last_level = int(lvl[r[0]])
            seen += r.tolist()
        want = np.nonzero(lvl >= 2)[0]
        assert sorted(seen) == want.tolist()


def test_header_is_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/dvbs2gpu.h must compile as C99 on its own (no C++, no torch types)"""
    import subprocess
    src = tmp_path / 'hdr.c'
    src.write_text('#include "dvbs2gpu.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', inc, '-c', str(src), '-o', str(tmp_path / 'hdr.o')],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _find_join(pkg, tail, bits):
    lib = pkg.load_library()
    tail, bits = np.ascontiguousarray(tail, np.uint8), np.ascontiguousarray(bits, np.uint8)
    inv = C.c_int(-1)
    at = lib.dvbs2gpu_dvbs_segrx_find_join(C.c_void_p(tail.ctypes.data), tail.size, C.c_void_p(bits.ctypes.data), bits.size, C.byref(inv))
    return int(at), inv.value


def test_dvbs_segment_join_rule(pkg):
    """host logic of the DVB-S segment receiver (csrc/dvbs_segrx.hip find_continuation): where a segment's decoded bits continue the
    stream already handed out -- exact, inverted (QPSK's 180-degree ambiguity), with bit errors inside the window and inside a key"""
    rng = np.random.default_rng(7)
    stream = rng.integers(0, 2, 40000, dtype=np.uint8)
    tail = stream[:20000]                                   # handed out so far
    seg = stream[20000 - 7000:]                             # the next segment started 7000 bits earlier
    assert _find_join(pkg, tail, seg) == (7000, 0)
    assert _find_join(pkg, tail, seg ^ 1) == (7000, 1)
    assert _find_join(pkg, tail[-300:], seg) == (7000, 0)   # 256 bits of tail are enough ...
    assert _find_join(pkg, tail[-255:], seg)[0] == -1       # ... fewer are not
    # the segment's unsettled start is garbage: irrelevant
    s2 = seg.copy(); s2[:5000] = rng.integers(0, 2, 5000)
    assert _find_join(pkg, tail, s2) == (7000, 0)
    # six bit errors inside the 256-bit window but outside the last key
    s3 = seg.copy(); s3[7000 - 256 + np.array([3, 40, 77, 120, 150, 180])] ^= 1
    assert _find_join(pkg, tail, s3) == (7000, 0)
    s3[7000 - 256 + 10] ^= 1                                # the seventh breaks it
    assert _find_join(pkg, tail, s3)[0] == -1
    # an error inside the last 64-bit key: the second key (64 bits earlier) anchors the match
    s4 = seg.copy(); s4[7000 - 5] ^= 1
    assert _find_join(pkg, tail, s4) == (7000, 0)
    s4[7000 - 70] ^= 1                                      # ... and the third
    assert _find_join(pkg, tail, s4) == (7000, 0)
    s4[7000 - 130] ^= 1                                     # all three keys hit
    assert _find_join(pkg, tail, s4)[0] == -1
    # unrelated bits, short inputs, bad arguments
    assert _find_join(pkg, tail, rng.integers(0, 2, 30000, dtype=np.uint8))[0] == -1
    assert _find_join(pkg, tail, seg[:200])[0] == -1
    assert _find_join(pkg, tail, seg[:7000]) == (7000, 0)   # the match may end exactly at the end of the segment
    assert _find_join(pkg, tail, seg[:6999])[0] == -1
    lib = pkg.load_library()
    assert lib.dvbs2gpu_dvbs_segrx_find_join(None, 10, None, 10, None) == pkg.ERR_ARG


def _reference_rows(rate, short):
    """the reference's expansion of the code (oracle): per row the information-bit positions, in its sequential row order"""
    p = orc.fec_params(rate, short)
    R = p['N'] - p['K']
    lib_o = orc.lib()
    lib_o.orc_ldpc_rows.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cnl = lib_o.orc_ldpc_rows(rate, short, None, None)
    pos = np.zeros((R, cnl), np.uint16)
    cn = np.zeros(R, np.uint8)
    lib_o.orc_ldpc_rows(rate, short, pos.ctypes.data, cn.ctypes.data)
    return p, R, pos, cn


@pytest.mark.parametrize('rate,short', [c for c in orc.ALL_CODES if c[1]])
def test_ldpc_wave_plan_keeps_the_reference_row_order(pkg, rate, short):
    """wave-per-frame decoder (csrc/ldpc_wave_plan.h): every row of a sweep sits in exactly one step, layers in order; the lane constants
    address exactly the reference's bits (own parity, previous parity, information links); rows of one step share no bit, and of two rows
    of a layer that share a bit the one the reference processes first sits in an EARLIER step"""
    lib = pkg.load_library()
    cnt = (C.c_int32 * 6)()
    assert lib.dvbs2gpu_ldpc_wave_plan_dump(rate, short, None, None, None, cnt) == 0
    lw, nl_min, nsteps, absent_base, nlanec, nst = list(cnt)
    p, R, pos, cn = _reference_rows(rate, short)
    q, K = R // 360, p['K']
    lanec = np.zeros(nlanec, np.uint32); steps = np.zeros(nst, np.uint16); layer_end = np.zeros(q, np.uint32)
    assert lib.dvbs2gpu_ldpc_wave_plan_dump(rate, short, lanec.ctypes.data, steps.ctypes.data, layer_end.ctypes.data, cnt) == 0
    assert nsteps % 4 == 0 and nst == (nsteps + 8) * 8 and np.all(steps[nsteps * 8:] == 0xffff)
    st = steps[:nsteps * 8].reshape(nsteps, 8).astype(np.int64)
    seen = np.zeros(R, np.int64)
    step_of = np.full(R, -1, np.int64)
    lo = 0
    for i in range(q):
        hi = int(layer_end[i]) * 4
        blk = st[lo:hi]
        rows = blk[blk != 0xffff]
        assert np.all(rows // 360 == i)                                   # a layer's steps hold that layer's rows only
        np.add.at(seen, rows, 1)
        for s_ in range(lo, hi):
            for rr in st[s_][st[s_] != 0xffff]:
                step_of[rr] = s_
        lo = hi
    assert lo == nsteps and np.all(seen == 1)
    # the bits every (row, slot) touches
    absent = lanec[absent_base:absent_base + q * 8].reshape(q, 8)
    c = lanec[:q * 8 * lw].reshape(q, 8, lw)
    thr, cA = (c & 0xffff).astype(np.int64), (c >> 16).astype(np.int64)
    assert nl_min == int(cn.min()) + 2
    for i in range(q):
        deg = int(cn[360 * i])
        touch = {}                                                          # bit -> rows of this layer
        for j in range(360):
            got = []
            for l8 in range(8):
                for kk in range(lw):
                    k = 8 * kk + l8
                    ab = (absent[i, l8] >> kk) & 1
                    assert ab == (k >= deg + 2)
                    if ab:
                        continue
                    a = j + (cA[i, l8, kk] - 360 if j >= thr[i, l8, kk] else cA[i, l8, kk])
                    if k == 0:
                        assert a == K + 360 * i + j
                    elif k == 1:
                        if i == 0 and j == 0:
                            continue                                        # (masked in the kernel: row 0 of layer 0 has no previous parity bit)
                        assert a == (K + 360 * (i - 1) + j if i else K + 360 * (q - 1) + j - 1)
                    else:
                        got.append(a)
            assert sorted(got) == sorted(int(x) for x in pos[360 * i + j, :deg])
            for b in got:
                touch.setdefault(b, []).append(j)
        for b, lst in touch.items():
            for a_, b_ in zip(lst, lst[1:]):                                # ascending row index = the reference's order
                assert step_of[360 * i + a_] < step_of[360 * i + b_]


@pytest.mark.parametrize('rate,short', orc.ALL_CODES)
def test_ldpc_address_table_holds_the_links_of_every_row(pkg, rate, short):
    """lane-per-row decoder: the per-code address table (regular codes of degree 2, 8, 12) names, per layer and row, exactly the reference's bits"""
    lib = pkg.load_library()
    cnt = (C.c_int32 * 2)()
    assert lib.dvbs2gpu_ldpc_addr_table_dump(rate, short, None, cnt) == 0
    n, stride = list(cnt)
    p, R, pos, cn = _reference_rows(rate, short)
    q = R // 360
    deg = int(cn.max())
    if n == 0:
        assert deg not in (2, 8, 12) or int(cn.min()) != deg
        return
    assert int(cn.min()) == deg and n == q * 384 * stride
    tab = np.zeros(n, np.uint32)
    assert lib.dvbs2gpu_ldpc_addr_table_dump(rate, short, tab.ctypes.data, cnt) == 0
    tab = tab.reshape(q, 384, stride)
    npi = (deg + 1) // 2
    a = np.stack([(tab[:, :360, k // 2] >> (16 * (k & 1))) & 0xffff for k in range(deg)], axis=-1).astype(np.int64)     # [layer][row][link]
    want = pos[:, :deg].astype(np.int64).reshape(q, 360, deg)
    assert np.array_equal(np.sort(a, axis=-1), np.sort(want, axis=-1))
    assert np.all(tab[:, 360:, :] == 0) and np.all(tab[:, :, npi:] == 0)


@pytest.mark.parametrize('rate,short', [(r, s) for r, s in orc.ALL_CODES if not s])
def test_ldpc_split_plan_keeps_the_reference_row_order(pkg, rate, short):
    """the half-row decoder's plan (csrc/ldpc_split_plan.h): every row of every layer appears
    exactly once, its table entry names exactly the bits the reference's row touches (information bits, own and previous parity bit), and two rows of a layer that
    share a bit run in the reference's order -- separated by a barrier (different pseudo-layers) unless the pseudo-layer resolves shared links itself (kinds 1, 8:
    test_ldpc_split_plan_side_entries below); every pseudo-layer runs all twelve waves"""
    sp = pkg.ldpc_split_plan(rate, short)
    p = orc.fec_params(rate, short)
    N, K = p['N'], p['K']
    R = N - K
    q = R // 360
    lib_o = orc.lib()
    lib_o.orc_ldpc_rows.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cnl = lib_o.orc_ldpc_rows(rate, short, None, None)
    if sp is None:
        why = pkg.load_library().dvbs2gpu_last_error().decode()
        assert 'does not take this code' in why and (rate, short) not in ((0, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0)), why
        return      # (layers with more than four shared links, more slots than a table entry holds, ...: the lane-per-row decoder keeps the code)
    pos = np.zeros((R, cnl), np.uint16); cn = np.zeros(R, np.uint8)
    lib_o.orc_ldpc_rows(rate, short, pos.ctypes.data, cn.ctypes.data)
    hs = sp['hs']
    assert 2 * hs in (cnl + 2, cnl + 3)            # (an odd number of links per row: half 1 carries a neutral slot)
    seen = np.zeros((q, 360), int)
    last_touch = {}            # bit -> (pseudo-layer, layer, row) of its latest toucher, in execution order
    for pl in range(sp['npl']):
        i = int(sp['layer'][pl]); kind = int(sp['kind'][pl])
        tab = sp['table'][pl]
        rows_here = []
        for pr in range(384):
            j = int(sp['row_of'][pl, pr])
            if j < 0:
                for h in (0, 1):       # idle lanes: scratch bytes only
                    a = [(int(tab[2 * pr + h, s >> 1]) >> (16 * (s & 1))) & 0xffff for s in range(hs)]
                    assert all(N <= x < N + 64 for x in a)
                continue
            assert int(sp['nw'][pl]) == 12
            seen[i, j] += 1
            addrs = []
            for h in (0, 1):
                addrs += [(int(tab[2 * pr + h, s >> 1]) >> (16 * (s & 1))) & 0xffff for s in range(hs)]
            want = [int(x) for x in pos[360 * i + j, :cn[360 * i + j]]] + [K + 360 * i + j]
            if i > 0: want.append(K + 360 * (i - 1) + j)
            elif j > 0: want.append(K + 360 * (q - 1) + j - 1)
            got = [a for a in addrs if a < N]
            assert sorted(got) == sorted(want), (pl, i, j)
            odd = 2 * hs - (cnl + 2)               # 1: half 1's last slot is the neutral link (a scratch byte)
            assert len(got) == len(addrs) - odd or (i == 0 and j == 0 and sp['noprev'][pl] and len(got) == len(addrs) - odd - 1)
            rows_here.append((j, got))
        within = {}
        for j, bits in rows_here:
            for b in bits:
                if b in last_touch:
                    pi, pj = last_touch[b]
                    assert (pi, pj) < (i, j), (b, pi, pj, i, j)                      # the reference's order across pseudo-layers (a barrier lies between)
                within.setdefault(b, []).append(j)
        for b, js in within.items():
            if len(js) > 1:                                                          # rows of ONE pseudo-layer share a bit: only where it orders shared links itself
                assert kind in (1, 8), (pl, i, js)
        for j, bits in rows_here:
            for b in bits: last_touch[b] = (i, j)
    assert (seen == 1).all()


@pytest.mark.parametrize('rate', [6, 3, 4, 5])
def test_ldpc_split_plan_side_entries(pkg, rate):
    """the speculative passes (kind 8, csrc/ldpc_split_kernel.hip: spec_layer) read a shared slot from the output cell of the row that touched the bit last before this
    row -- in the reference's row order (layered_decoder.hh:46-74) -- or from the bit itself where that row has level 1 or does not exist; the plan's side entries must say
    exactly that, a row's level must be one more than its deepest predecessor's, and only slots 0..3 of half 0 may be shared"""
    sp = pkg.ldpc_split_plan(rate, False)
    hs, N = sp['hs'], orc.fec_params(rate, False)['N']
    seen8 = 0
    for pl in range(sp['npl']):
        if int(sp['kind'][pl]) != 8:
            continue
        seen8 += 1
        tab = sp['table'][pl]
        side = sp['words'][int(sp['ent_off'][pl]):int(sp['ent_off'][pl]) + 2 * 384].reshape(384, 2)
        depth = int(sp['aux'][pl]) >> 16
        last = {}; level = {}
        succ_want = {}
        for j in range(360):
            assert int(sp['row_of'][pl, j]) == j
            addrs = []
            for h in (0, 1):
                addrs += [(int(tab[2 * j + h, s >> 1]) >> (16 * (s & 1))) & 0xffff for s in range(hs)]
            rw = [(int(tab[2 * j + h, hs >> 1]) >> (16 * (hs & 1))) & 0xffff for h in (0, 1)]
            lvl = 1
            for k, a in enumerate(addrs):
                if a >= N:
                    continue              # (a neutral slot's scratch byte)
                f = (int(side[j, k >> 1]) >> (16 * (k & 1))) & 0x7ff if k < 4 else 0
                if a in last:
                    assert k < 4, (pl, j, k)
                    pj, pk = last[a]
                    succ_want[(pj, pk)] = True
                    lvl = max(lvl, level[pj] + 1)
                    assert f == (4 * (j - pj) - pk if level[pj] > 1 else 0), (pl, j, k)
                else:
                    assert f == 0, (pl, j, k)
                last[a] = (j, k)
            level[j] = lvl
            assert rw[0] & 0xff == lvl and rw[1] & 0xff == lvl, (pl, j)
        assert max(level.values()) == depth
        for j in range(360):
            for k in range(4):
                bit = (int(side[j, k >> 1]) >> (16 * (k & 1) + 15)) & 1
                assert bit == int(succ_want.get((j, k), False)), (pl, j, k)
        assert not side[360:].any()
    assert seen8 >= 1 or rate == 3          # (rate 1/2: its one layer with shared bits is a chain of 32 steps -- the chain walk's)


def test_fleet_without_a_gpu_fails_loudly(pkg):
    """the fleet has no CPU fallback either: without a HIP device dvbs2gpu_fleet_create returns the no-device error (and dvbs2gpu_device_count says 0); the placement rule
    alone (dvbs2gpu_fleet_plan) needs no device"""
    import torch
    assert pkg.fleet_plan([4, 4, 14, 14], [1.0, 1.0, 3.0, 3.0], 2, 1.0) in ([1, 1, 0, 0], [0, 0, 1, 1])
    if torch.cuda.is_available():
        pytest.skip('GPU present: the no-device path cannot be shown')
    assert pkg.load_library().dvbs2gpu_device_count() == 0
    with pytest.raises(pkg.Dvbs2GpuError) as e:
        pkg.Fleet([0])
    assert 'no CPU fallback' in str(e.value) or 'no HIP device' in str(e.value)
