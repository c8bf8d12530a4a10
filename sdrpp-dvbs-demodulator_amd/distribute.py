"""Work distribution over the GPUs of one node (the reference runs one independent DVBS2Demod per transponder,
src/main.cpp:588,595, so transponders are the unit that shards).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Nothing is
exchanged inside a frame or a stream; the collectives are the ones work DISTRIBUTION needs:

  * broadcast (root 0) of the transponder table and the engine configuration, so that every rank derives the same assignment
    and the same demodulator settings from one source of truth;
  * gather of the decoded BBFRAMEs and their per-frame statistics to the egress rank (the rank that feeds the UDP sink in the
    plugin's terms, main.cpp:532-558), which puts them back into the input order of the transponders.

Assignment: transponders are weighted (symbol rate x LDPC work per symbol) and placed by longest-processing-time first, with
equal-MODCOD transponders kept on the same rank while the balance allows it -- LDPC batches then stay homogeneous per GPU."""
import pickle


def shard_range(n_units, rank, world):
    """contiguous balanced partition of range(n_units): (start, stop) of `rank`"""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_by_weight(weights, world):
    """greedy longest-processing-time assignment of weighted units to `world` ranks: list of unit-index lists"""
    order = sorted(range(len(weights)), key=lambda i: -weights[i])
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        out[r].append(i)
        loads[r] += weights[i]
    return out


def assign_transponders(table, world, tolerance=0.25):
    """table: list of dicts with 'modcod' and 'weight'.  Returns per-rank lists of table indices (ascending).
    First choice: whole MODCOD groups placed by longest-processing-time first (every MODCOD on exactly one rank: homogeneous LDPC
    batches per GPU), accepted when no rank ends up more than `tolerance` above the average load.  Otherwise the MODCOD-sorted
    list is cut into `world` contiguous pieces of near-equal weight: only the groups at the cuts are shared by two ranks."""
    n = len(table)
    if world <= 1 or n == 0:
        return [list(range(n))] + [[] for _ in range(max(world, 1) - 1)]
    groups = {}
    for i, t in enumerate(table):
        groups.setdefault(t['modcod'], []).append(i)
    total = sum(t['weight'] for t in table) or 1.0
    avg = total / world
    keys = sorted(groups, key=lambda m: (-sum(table[i]['weight'] for i in groups[m]), m))
    if len(keys) >= world:
        placed = shard_by_weight([sum(table[i]['weight'] for i in groups[m]) for m in keys], world)
        out = [sorted(i for g in part for i in groups[keys[g]]) for part in placed]
        if max(sum(table[i]['weight'] for i in x) for x in out) <= (1.0 + tolerance) * avg:
            return out
    order = [i for m in sorted(groups) for i in groups[m]]
    out = [[] for _ in range(world)]
    acc, r = 0.0, 0
    for pos, i in enumerate(order):
        left = len(order) - pos
        # move to the next rank when this one has its share (and leave at least one unit for every remaining rank)
        if r < world - 1 and out[r] and (acc + 0.5 * table[i]['weight'] > (r + 1) * avg or left <= world - 1 - r):
            r += 1
        out[r].append(i)
        acc += table[i]['weight']
    return [sorted(x) for x in out]


class Distributor:
    """Broadcast / gather plumbing of one rank.  `dist` is torch.distributed (already initialised) or None for a single process;
    `device` is where collective buffers live ('cuda:k' with nccl, 'cpu' with gloo)."""

    def __init__(self, dist, device, egress=0):
        self.dist, self.device, self.egress = dist, device, egress
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    # ---- configuration / tables: root -> everyone
    def broadcast_object(self, obj, src=0):
        """a picklable object (transponder table, configuration) from `src` to every rank, as a length + byte tensor broadcast"""
        import torch
        if self.dist is None:
            return obj
        if self.rank == src:
            raw = pickle.dumps(obj)
            n = torch.tensor([len(raw)], dtype=torch.int64, device=self.device)
        else:
            n = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.dist.broadcast(n, src)
        if self.rank == src:
            buf = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        else:
            buf = torch.empty(int(n.item()), dtype=torch.uint8, device=self.device)
        self.dist.broadcast(buf, src)
        return obj if self.rank == src else pickle.loads(buf.cpu().numpy().tobytes())

    def broadcast_tensor(self, t, src=0):
        """a table that lives in device memory (same shape and dtype on every rank), in place"""
        if self.dist is not None:
            self.dist.broadcast(t, src)
        return t

    # ---- results: everyone -> egress rank, back into input order
    def gather_units(self, local_ids, payload, counts, n_units):
        """local_ids: the global unit (transponder) index of each local row; payload: uint8 tensor [n_local, width] on
        self.device; counts: int32 tensor [n_local] (valid bytes per row).  On the egress rank returns (payload [n_units, width],
        counts [n_units]) with row u = unit u; elsewhere (None, None).  Rows are padded to the largest per-rank count so that one
        fixed-size gather carries everything; a second one carries ids + counts."""
        import torch
        payload = payload.to(self.device)            # (gloo: collective buffers live on the host)
        counts = counts.to(self.device)
        width = int(payload.shape[1]) if payload.dim() == 2 else 0
        if self.dist is None:
            out = torch.zeros((n_units, width), dtype=torch.uint8, device=self.device)
            cnt = torch.zeros(n_units, dtype=torch.int32, device=self.device)
            idx = torch.as_tensor(local_ids, dtype=torch.int64, device=self.device)
            out[idx] = payload
            cnt[idx] = counts
            return out, cnt
        # one all-reduce agrees on the row count AND the row width: a rank without local units may have passed an empty / 1-D payload.  The
        # verdict on the widths is COLLECTIVE (the smallest width of a rank that has rows travels as a negated maximum): every rank raises, or
        # none -- a rank that raised alone would leave the others waiting in the gathers below
        big = 1 << 40
        nmax = torch.tensor([len(local_ids), width if len(local_ids) else 0, -(width if len(local_ids) else big)], dtype=torch.int64, device=self.device)      # (only ranks with rows vote on the width)
        self.dist.all_reduce(nmax, op=self.dist.ReduceOp.MAX)
        nmax, wmax, wmin = int(nmax[0].item()), int(nmax[1].item()), -int(nmax[2].item())
        if wmin != big and wmin != wmax:
            raise ValueError('gather_units: payload rows are %d bytes wide on one rank and %d on another (%d here)' % (wmin, wmax, width))
        width = wmax
        pad = torch.zeros((nmax, width), dtype=torch.uint8, device=self.device)
        meta = torch.full((nmax, 2), -1, dtype=torch.int32, device=self.device)
        k = len(local_ids)
        if k:
            pad[:k] = payload
            meta[:k, 0] = torch.as_tensor(local_ids, dtype=torch.int32, device=self.device)
            meta[:k, 1] = counts
        if self.rank == self.egress:
            gp = [torch.empty_like(pad) for _ in range(self.world)]
            gm = [torch.empty_like(meta) for _ in range(self.world)]
        else:
            gp = gm = None
        self.dist.gather(pad, gp, dst=self.egress)
        self.dist.gather(meta, gm, dst=self.egress)
        if self.rank != self.egress:
            return None, None
        allp, allm = torch.cat(gp), torch.cat(gm)
        keep = allm[:, 0] >= 0
        ids = allm[keep, 0].to(torch.int64)
        out = torch.zeros((n_units, width), dtype=torch.uint8, device=self.device)
        cnt = torch.zeros(n_units, dtype=torch.int32, device=self.device)
        out[ids] = allp[keep]
        cnt[ids] = allm[keep, 1]
        return out, cnt

    def max_over_ranks(self, x):
        import torch
        if self.dist is None:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_over_ranks(self, x):
        """every rank's value of a scalar, in rank order (one all-gather): per-rank step times next to the per-rank loads"""
        import torch
        if self.dist is None:
            return [float(x)]
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(v.item()) for v in out]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
