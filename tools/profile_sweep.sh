#!/bin/bash
# rocprofv3 passes over the finest-level smoother alone (tools/relax_only.py: 6 x Relax(2) at 513^3 fp64), one launch per
# red+black sweep (relax3d.fused=1) next to one launch per colour:   bash tools/profile_sweep.sh r03_sweep
set -e
P=${1:-rNN_sweep}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$P
mkdir -p $O
export TMPDIR=/tmp
for F in 1 0; do
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$F -- python3 $R/tools/relax_only.py relax3d.fused=$F > $O/stats$F.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch$F -- python3 $R/tools/relax_only.py relax3d.fused=$F > $O/fetch$F.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write$F -- python3 $R/tools/relax_only.py relax3d.fused=$F > $O/write$F.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/sq$F -- python3 $R/tools/relax_only.py relax3d.fused=$F > $O/sq$F.log 2>&1
cd $R
python tools/pmc_summary.py $O/stats$F $O/fetch$F $O/write$F $O/sq$F --out gpurun_out/${P}_fused$F
rm -rf $O/stats$F $O/fetch$F $O/write$F $O/sq$F
done
