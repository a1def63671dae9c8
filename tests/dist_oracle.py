"""How the CPU oracle is configured for the global (assembled) hierarchy of a DistributedAMG -- test infrastructure."""


def oracle_sm_types(amg):
    """per-level smoother names for oracle.pyoracle.Oracle over amg.global_levels()"""
    if amg.sm_type == "jacobi":
        return ["jacobi"] * (amg.k + amg.tail_hier.n_levels)
    if amg.sm_type == "bgs":
        return ["bgs_mc"] * (amg.k + amg.tail_hier.n_levels)
    # gs / hgs: explicit visiting order (+ blocks with frozen couplings) on the distributed levels; tail levels that sweep in
    # the block-hybrid form carry their order too, the others run in colour order
    return ["gs_order"] * amg.k + ["gs_order" if getattr(L, "gs_order", None) is not None else "gs_mc" for L in amg.tail_hier.levels]


def oracle_bgs(amg, levels):
    """bgs=... argument for the oracle over amg.global_levels() (block smoother only)"""
    return [getattr(L, "bgs", None) for L in levels] if amg.sm_type == "bgs" else None
