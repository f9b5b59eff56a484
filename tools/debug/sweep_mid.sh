python tools/debug/relax_time.py 257 f64
for s in 1242 1422 1442 1184 1424; do for z in 0 16 32 64; do MGX_PARAMS=relax3d.lds=$s,relax3d.zchunk=$z python tools/debug/relax_time.py 257 f64; done; done
for z in 8 12 24 32; do MGX_PARAMS=relax3d.zchunk=$z python tools/debug/relax_time.py 257 f64; done
