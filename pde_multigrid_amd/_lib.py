"""ctypes loader of pde_multigrid_amd/lib/libmgx.so (HIP kernels + C-ABI + C host layer).

The product path has NO CPU fallback for the 2D/3D operators: if the library is missing this
module raises ImportError, and every call that needs a device returns MGX_ERR_NOGPU
(raised as MgxError) when no MI355X is visible.

Load order: PyTorch-ROCm bundles its own copies of libamdhip64 / librccl (same sonames as /opt/rocm's).
A process that uses both must import torch BEFORE this module, so that libmgx binds to the copies torch
loaded; the opposite order mixes two ROCm runtimes and aborts at exit.  Only the gloo test workers (CPU emulation of
the slab schedule) do that; bench.py's ranks and everything else never import torch (pde_multigrid_amd/launch.py is the
control plane of N > 1 runs), so the data plane runs on the ROCm runtime and RCCL libmgx was built with.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# tools/ may point this at the diagnostic twin (lib/libmgx_diag.so, `make -C pde_multigrid_amd/csrc diag`)
LIB_PATH = os.environ.get("MGX_LIB_PATH") or os.path.join(_HERE, "lib", "libmgx.so")

MGX_OK, MGX_ERR_INVALID, MGX_ERR_SIZE, MGX_ERR_HIP, MGX_ERR_NOMEM, MGX_ERR_RCCL, MGX_ERR_NOGPU = range(7)
REF_COMPAT, CORRECT = 0, 1
UNIQUE_ID_BYTES = 128


class MgxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (status_string(status), message))
        self.status = status


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "pde_multigrid_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C pde_multigrid_amd/csrc`; there is no CPU fallback for the HIP path" % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
lib.mgx_status_string.restype = C.c_char_p
lib.mgx_last_error.restype = C.c_char_p
lib.mgx_version.restype = C.c_char_p
lib.mgx_ctx_last_relax_kernel.restype = C.c_char_p
lib.mgx_ctx_last_rr_kernel.restype = C.c_char_p
lib.mgx_ctx_last_corr_kernel.restype = C.c_char_p


def status_string(status):
    return lib.mgx_status_string(int(status)).decode()


def check(status):
    if status != MGX_OK:
        raise MgxError(status, lib.mgx_last_error().decode())
    return status
