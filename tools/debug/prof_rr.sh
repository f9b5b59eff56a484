set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for x in 0 1; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/rr_fetch_$x -- python3 $R/tools/rr_only.py residual_restrict3d.xcd=$x >> $R/gpurun_out/prof_rr.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/rr_write_$x -- python3 $R/tools/rr_only.py residual_restrict3d.xcd=$x >> $R/gpurun_out/prof_rr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rr_stats_$x -- python3 $R/tools/rr_only.py residual_restrict3d.xcd=$x >> $R/gpurun_out/prof_rr.log 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/rr_stats_$x gpurun_out/rr_fetch_$x gpurun_out/rr_write_$x --out gpurun_out/rr_x$x; cd /tmp
done
