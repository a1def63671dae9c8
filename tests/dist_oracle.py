"""How the CPU oracle is configured for the global (assembled) hierarchy of a DistributedAMG -- test infrastructure."""


def oracle_sm_types(amg):
    """per-level smoother names for oracle.pyoracle.Oracle over amg.global_levels()"""
    if amg.sm_type == "jacobi":
        return ["jacobi"] * (amg.k + amg.tail_hier.n_levels)
    if amg.sm_type == "bgs":
        return ["bgs_mc"] * (amg.k + amg.tail_hier.n_levels)
    return ["gs_order"] * amg.k + ["gs_mc"] * amg.tail_hier.n_levels          # gs and hgs: explicit order (+ blocks) on the distributed levels


def oracle_bgs(amg, levels):
    """bgs=... argument for the oracle over amg.global_levels() (block smoother only)"""
    return [getattr(L, "bgs", None) for L in levels] if amg.sm_type == "bgs" else None
