"""Unit sharding across the GPUs of one node.

Units are independent, so a batch shards by unit with no data-path collective:
rank r decodes one contiguous range of units, balanced by bytes, and owns one
contiguous span of the output.  The only exchange is a host-side gather of
sizes/status in unit order (first error wins), done here with
torch.distributed object collectives; the bytes themselves never cross ranks.
"""


def partition(weights, world):
    """Contiguous ranges [lo, hi) per rank with near-equal weight sums (weights = bytes per unit)."""
    n = len(weights)
    total = sum(weights)
    bounds, acc, k = [0], 0, 1
    for i, w in enumerate(weights):
        acc += w
        while k < world and acc * world >= total * k and len(bounds) <= k:
            bounds.append(i + 1)
            k += 1
    while len(bounds) < world:
        bounds.append(n)
    bounds.append(n)
    return [(bounds[r], max(bounds[r], bounds[r + 1])) for r in range(world)]


def shard_units(units, rank, world, by="in"):
    """This rank's contiguous share of `units` (objects with in_off/in_limit/out_cap)."""
    if by == "out":
        w = [max(1, int(u.out_cap)) for u in units]
    else:
        offs = [int(u.in_off) for u in units] + [int(units[-1].in_limit) if units else 0]
        w = [max(1, offs[i + 1] - offs[i]) for i in range(len(units))]
    lo, hi = partition(w, world)[rank]
    return lo, hi


def gather_outcomes(local, dist=None):
    """All ranks' per-unit (status, out_len) lists concatenated in rank (= unit) order, and the first
    failing unit's global index (or None).  `local` is this rank's list."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        parts = [local]
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, local)
    flat = [x for p in parts for x in p]
    first_bad = next((i for i, (st, _) in enumerate(flat) if st != 0), None)
    return flat, first_bad
