for i in 1 2 3; do for lib in libmgx_prev.so libmgx.so; do echo -n "$lib "; MGX_LIB_PATH=$GRAFT_REPO_ROOT/pde_multigrid_amd/lib/$lib python tools/debug/rr_time.py ${1:-513} ${2:-f64}; done; done
