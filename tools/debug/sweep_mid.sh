python tools/debug/relax_time.py 129 f64
for z in 0 4 8 16 32; do MGX_PARAMS=relax3d.lds=1184,relax3d.zchunk=$z python tools/debug/relax_time.py 129 f64; done
for z in 0 8 16; do MGX_PARAMS=relax3d.lds=184,relax3d.zchunk=$z python tools/debug/relax_time.py 129 f64; done
for p in relax3d.zchunk=1 relax3d.zchunk=2 relax3d.zchunk=8 relax3d.ty=2 relax3d.ty=8 relax3d.rows=2 relax3d.rows=1 relax3d.ty=8,relax3d.rows=2; do MGX_PARAMS=$p python tools/debug/relax_time.py 129 f64; done
python tools/debug/relax_time.py 65 f64
for p in relax3d.zchunk=1 relax3d.zchunk=2 relax3d.ty=2 relax3d.rows=2 relax3d.ty=2,relax3d.rows=2; do MGX_PARAMS=$p python tools/debug/relax_time.py 65 f64; done
