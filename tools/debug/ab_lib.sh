# same-box A/B of two builds of the library: bash tools/debug/ab_lib.sh  (pde_multigrid_amd/lib/libmgx_prev.so vs libmgx.so)
for i in 1 2 3; do
for lib in libmgx_prev.so libmgx.so; do
MGX_LIB_PATH=$GRAFT_REPO_ROOT/pde_multigrid_amd/lib/$lib python bench.py --no-cpu-baseline --steps 40 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done; done
