for p in "" "residual_restrict3d.stream=2,residual_restrict3d.tyw=8" "residual_restrict3d.stream=2,residual_restrict3d.tyw=8,residual_restrict3d.pzchunk=32" "residual_restrict3d.stream=2,residual_restrict3d.tyw=8,residual_restrict3d.pzchunk=64" "residual_restrict3d.stream=2,residual_restrict3d.tyw=8,residual_restrict3d.pzchunk=16" "residual_restrict3d.rows=4" "residual_restrict3d.pzchunk=43" "residual_restrict3d.pzchunk=128" ""; do
  MGX_PARAMS=$p timeout -k 10 60 python tools/probes/rr_time.py 513 f32 || exit 1
done
