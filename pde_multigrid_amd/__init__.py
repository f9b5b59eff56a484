"""pde_multigrid_amd -- MI355X-native geometric multigrid (V-cycle / FMG) behind the operator
surface of MisterPup/PDE-MultiGrid's NOCUDA_TESI classes.

    C-ABI over the HIP kernels ............ include/mgx.h          (csrc/*.hip)
    C host layer mirroring the reference .. include/mg_multigrid.h (csrc/host/*.c)
    this package ........................... ctypes views of both, for tests and bench.py
"""
from ._lib import (CORRECT, LIB_PATH, MGX_ERR_INVALID, MGX_ERR_NOGPU, MGX_ERR_SIZE, MGX_OK, REF_COMPAT,  # noqa: F401
                   MgxError, check, lib, status_string)
from .multigrid import (Context, DistMultiGrid3D, LocalGroup, dist_num_levels, slab_plan, MultiGrid1D, MultiGrid2D, MultiGrid3D, coarse_size, grid_spacing,  # noqa: F401
                        num_grids, ops2d, ops3d, ops3dxs, solve1d, solve2d, solve3d, solve3d_from_zero, xs_geometry, xs_pack, xs_unpack)

__all__ = ["Context", "DistMultiGrid3D", "LocalGroup", "dist_num_levels", "slab_plan", "MultiGrid1D", "MultiGrid2D", "MultiGrid3D", "ops2d", "ops3d", "ops3dxs", "xs_pack", "xs_unpack", "solve1d", "solve2d", "solve3d", "solve3d_from_zero",
           "num_grids", "coarse_size", "grid_spacing", "MgxError", "REF_COMPAT", "CORRECT", "lib", "check"]
