python tools/debug/relax_time.py 257 f32
for s in 1282 1242 1442 1422; do for z in 8 16 32; do MGX_PARAMS=relax3d.lds=$s,relax3d.zchunk=$z python tools/debug/relax_time.py 257 f32; done; done
