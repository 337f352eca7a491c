"""How the ranks of one chain split the amplitude-sampling matvec: A = 1 + S^1/2 sum_bands sum_rings (...)
S^1/2 is a double sum, so ranks may own a subset of the bands, a subset of the ring pairs, or both (hybrid).  Pure ring
sharding balances perfectly but shrinks the per-rank Legendre problem (256 ring pairs at Nside 1024 on 8 GPUs run at
64 % of the single-GPU efficiency, measured with tools/cr_time_rank.py); pure band sharding keeps the kernels large
but balances badly (9 bands on 8 GPUs).  ``plan_shards`` picks the factorisation world = band_parts x ring_parts with
the smallest estimated time."""

# measured on MI355X at the cfg3 geometry with the round-2 kernels (tools/cr_time_rank.py; one rank's
# matvec + invM: 7.23 ms alone, 4.01 / 2.38 / 1.41 ms as 1 of 2 / 4 / 8 ring sets with block ring ownership, 1.6 ms as 1
# of 16: 128 pairs leave the synthesis without its workgroup form): compute efficiency of one rank's share under
# ring_parts-way ring sharding
RING_EFF = {1: 1.0, 2: 0.90, 4: 0.76, 8: 0.64}


def _eff(r):
    if r in RING_EFF:
        return RING_EFF[r]
    return RING_EFF[8] * 8.0 / r if r > 8 else 1.0


def _band_penalty(nb):
    # cost per band relative to a rank that holds all nine (smaller map batches per launch: 3..5 maps take the DPP form
    # of the adjoint, 1..2 the VALU kernel, the synthesis runs one batch).  Measured per-rank shares: 2 x 4: 1.68 ms,
    # 2 x 8: 1.06 ms (5 bands: 1.19 / 1.30); 4 x 2: 2.04 ms, 4 x 4: 1.26 ms (3 bands: 1.44 / 1.48); fewer bands scaled
    return 1.0 if nb >= 9 else (1.25 if nb >= 5 else (1.46 if nb >= 3 else (1.55 if nb == 2 else 1.6)))


def plan_shards(nband, world):
    """-> (band_parts, ring_parts) minimising  max_bands_per_group / (ring_parts * eff(ring_parts)).  Pure ring sharding
    (one communicator, perfect balance) is kept unless a hybrid layout is estimated at least 5 % faster."""
    best = None
    for bp in range(1, world + 1):
        if world % bp or bp > nband:
            continue
        rp = world // bp
        mb = -(-nband // bp)                       # bands of the most loaded group
        cost = mb * (_band_penalty(mb) if bp > 1 else 1.0) / (rp * _eff(rp)) * (1.05 if bp > 1 else 1.0)
        if best is None or cost < best[0] - 1e-12:
            best = (cost, bp, rp)
    return best[1], best[2]


def rank_layout(nband, world, rank, band_parts=None, ring_parts=None):
    """-> dict(bands=[...], ring_index, ring_parts, ring_group=[ranks sharing these bands], band_parts)."""
    if band_parts is None or ring_parts is None:
        band_parts, ring_parts = plan_shards(nband, world)
    assert band_parts * ring_parts == world
    bg, ri = divmod(rank, ring_parts)
    lo = bg * nband // band_parts
    hi = (bg + 1) * nband // band_parts
    return dict(bands=list(range(lo, hi)), ring_index=ri, ring_parts=ring_parts, band_parts=band_parts,
                ring_group=[bg * ring_parts + i for i in range(ring_parts)])
