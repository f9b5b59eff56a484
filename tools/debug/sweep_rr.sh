for n in 257 129; do
MGX_PARAMS= python tools/debug/rr_time.py $n f64
for pz in 0 6 8 10 16; do MGX_PARAMS=residual_restrict3d.tyw=8,residual_restrict3d.pzchunk=$pz python tools/debug/rr_time.py $n f64; done
for pz in 4 6 8 10; do MGX_PARAMS=residual_restrict3d.pzchunk=$pz python tools/debug/rr_time.py $n f64; done
done
