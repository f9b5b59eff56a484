for i in 1 2; do
MGX_PARAMS= python tools/debug/rr_time.py 1025 f64
MGX_PARAMS=residual_restrict3d.pzchunk=171 python tools/debug/rr_time.py 1025 f64
MGX_PARAMS=residual_restrict3d.rows=4 python tools/debug/rr_time.py 1025 f64
done
