"""Work distribution over the GPUs of one node.  Transponder streams (and, inside a stream, FEC frames) are
independent units: each rank takes a contiguous, balanced slice; no data-path collective exists.
The only collectives used anywhere are the barrier / max-reduce of the benchmark and an optional gather of
per-rank frame counts for reporting."""


def shard_range(n_units, rank, world):
    """contiguous balanced partition of range(n_units): returns (start, stop) of `rank`"""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_by_weight(weights, world):
    """greedy longest-processing-time assignment of weighted units (e.g. symbol rate x bits/symbol of a
    transponder) to `world` ranks; returns a list of unit-index lists, heaviest units first"""
    order = sorted(range(len(weights)), key=lambda i: -weights[i])
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        out[r].append(i)
        loads[r] += weights[i]
    return out
