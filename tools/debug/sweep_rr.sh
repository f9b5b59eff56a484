for pz in 0 4 8 11 16 32; do MGX_PARAMS=residual_restrict3d.stream=2,residual_restrict3d.rows=2,residual_restrict3d.pzchunk=$pz python tools/debug/rr_time.py 257 f64; done
for pz in 0 4 8 16; do MGX_PARAMS=residual_restrict3d.stream=2,residual_restrict3d.rows=2,residual_restrict3d.pzchunk=$pz python tools/debug/rr_time.py 129 f64; done
for n in 65 33; do for p in residual_restrict3d.stream=3 residual_restrict3d.stream=2,residual_restrict3d.rows=2 residual_restrict3d.stream=2,residual_restrict3d.rows=2,residual_restrict3d.pzchunk=2 residual_restrict3d.stream=2,residual_restrict3d.rows=4,residual_restrict3d.tyw=4; do MGX_PARAMS=$p python tools/debug/rr_time.py $n f64; done; done
MGX_PARAMS=residual_restrict3d.stream=2,residual_restrict3d.rows=2 python tools/debug/rr_time.py 257 f32
MGX_PARAMS=residual_restrict3d.stream=3 python tools/debug/rr_time.py 257 f32
