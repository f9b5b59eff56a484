"""GPU suite: the drop-in boundary called from C, not through ctypes.  The program is INTEGRATION.md section A (the
replacement of the reference's driver, N3/Poisson3DSolver.cpp:6-51) at n = 17 with FullMultiGridVCycle(0, 1, 2, 2) in
fp32; it is compiled here with gcc -std=c11 against include/*.h and libmgx.so, run, and the FNV hash it prints of
grids3D[0]->h_v is compared with the known answer the compiled reference gave (tests/golden/known_answers.json,
3d_n17_fmg122).  A second program drives the raw operator ABI (mgx.h) the way INTEGRATION.md section B's stubs do."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
#include "mg_multigrid.h"
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
    float range[] = {0, 1, 0, 1, 0, 1};              /* N3/Poisson3DSolver.cpp:14 */
    int   finestGridSizeXYZ[] = {17, 17, 17};         /* :16-17 */
    int   v0 = 1, v1 = 2, v2 = 2;                    /* :18-20 */
    mgx_ctx* ctx;  mgMultiGrid3D_f32* multiGrid3D;
    if (mgx_ctx_create(0, &ctx) ||
        mgMultiGrid3D_f32_create(ctx, finestGridSizeXYZ, range, &multiGrid3D)) {   /* MultiGrid3D multiGrid3D(size, range)  :22 */
        fprintf(stderr, "%s\n", mgx_last_error());  return 1;
    }
    if (mgMultiGrid3D_f32_FullMultiGridVCycle(multiGrid3D, 0, v0, v1, v2)) {        /* :34 */
        fprintf(stderr, "%s\n", mgx_last_error());  return 1;
    }
    if (mgMultiGrid3D_f32_download_v(multiGrid3D, 0, NULL)) {                       /* fills grids3D[0]->h_v */
        fprintf(stderr, "%s\n", mgx_last_error());  return 1;
    }
    const float* v = multiGrid3D->grids3D[0]->h_v;
    uint64_t h = 0xcbf29ce484222325ull;             /* SURVEY.md section 8c: FNV over the 32-bit patterns in memory order */
    for (int i = 0; i < 17 * 17 * 17; i++) { uint32_t b; memcpy(&b, &v[i], 4); h = (h ^ b) * 0x100000001b3ull; }
    printf("numGrids %d\nhash %016llx\ncentre %.9g\n", multiGrid3D->numGrids, (unsigned long long)h, v[8 + 8 * 17 + 8 * 17 * 17]);
    mgMultiGrid3D_f32_destroy(multiGrid3D);  mgx_ctx_destroy(ctx);
    return 0;
}
"""

OPS = r"""
#include "mgx.h"
#include <stdio.h>
#include <stdlib.h>

/* MultiGrid3D::setToValue + Relax through the raw operator ABI, natural layout (INTEGRATION.md section B) */
int main(void) {
    enum { N = 9 };
    int n[3] = {N, N, N};
    float h[3] = {0.125f, 0.125f, 0.125f};
    float host[N * N * N], f[N * N * N];
    for (int i = 0; i < N * N * N; i++) { host[i] = 0.0f; f[i] = 1.0f; }
    mgx_ctx* ctx;  void *dv, *df;
    if (mgx_ctx_create(0, &ctx)) { fprintf(stderr, "%s\n", mgx_last_error()); return 1; }
    if (mgx_malloc(ctx, sizeof host, &dv) || mgx_malloc(ctx, sizeof f, &df) || mgx_memcpy_h2d(ctx, dv, host, sizeof host) ||
        mgx_memcpy_h2d(ctx, df, f, sizeof f) || mgx3d_relax_f32(ctx, (float*)dv, (const float*)df, n, h, 1) ||
        mgx_memcpy_d2h(ctx, host, dv, sizeof host)) { fprintf(stderr, "%s\n", mgx_last_error()); return 1; }
    /* red = (x + y + z) even: (2,1,1) from zeros is (0 - f h^6) / (6 h^4); black (1,1,1) then sees three red neighbours */
    printf("red %.9g\nblack %.9g\nboundary %.9g\n", host[2 + N + N * N], host[1 + N + N * N], host[0]);
    int bad = mgx3d_relax_f32(ctx, (float*)dv, (const float*)df, n, h, -1);   /* error path: status + message, no abort */
    printf("status %s\n", mgx_status_string(bad));
    mgx_free(ctx, dv); mgx_free(ctx, df); mgx_ctx_destroy(ctx);
    return 0;
}
"""


def _build_and_run(tmp_path, name, source):
    if not shutil.which("gcc"):
        pytest.skip("no gcc on this box")
    src = tmp_path / (name + ".c")
    src.write_text(source)
    exe = tmp_path / name
    lib = os.path.join(ROOT, "pde_multigrid_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I" + os.path.join(ROOT, "include"), str(src), "-L" + lib, "-lmgx",
                           "-Wl,-rpath," + lib, "-o", str(exe)])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return dict(line.split(" ", 1) for line in p.stdout.strip().splitlines())


def test_c_driver_fmg_n17_matches_reference_known_answer(tmp_path, known_answers):
    out = _build_and_run(tmp_path, "Poisson3D", DRIVER)
    ka = known_answers["3d_n17_fmg122"]
    assert out["numGrids"] == "4"
    assert out["hash"] == ka["hash"] == "600f607b6dcb83d9"
    import numpy as np
    assert np.float32(float(out["centre"])) == np.float32(ka["centre"])  # %.9g round-trips a float


def test_c_caller_of_the_operator_abi(tmp_path):
    out = _build_and_run(tmp_path, "ops", OPS)
    import numpy as np
    f32 = np.float32
    h2 = f32(0.125) * f32(0.125)
    c, fh = h2 * h2, ((f32(1.0) * h2) * h2) * h2
    den = f32(2) * ((c + c) + c)
    zero = f32(0)
    red = ((((((zero * c + zero * c) + zero * c) + zero * c) + zero * c) + zero * c) - fh) / den  # N3/MultiGrid3D.cpp:532
    black = ((((((zero * c + red * c) + zero * c) + red * c) + zero * c) + red * c) - fh) / den   # O, N, D are boundary zeros
    got = {k: f32(float(out[k])) for k in ("red", "black", "boundary")}
    assert got["red"] == red and got["black"] == black and got["boundary"] == 0.0
    assert out["status"] == "MGX_ERR_INVALID"
