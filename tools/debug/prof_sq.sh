set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/sq_pass -- python3 $R/bench.py --no-cpu-baseline --steps 5 >> $R/gpurun_out/prof_sq.log 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/sq_pass --out gpurun_out/sq
