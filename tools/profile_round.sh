#!/bin/bash
# The round's profile set, run on the GPU box:   bash tools/profile_round.sh r02_final
# rocprofv3 runs from /tmp with TMPDIR=/tmp, the interpreter directly after `--`; counters in passes of their own.
# Results go to gpurun_out/ (scratch); the summaries are copied into profiles/ by hand.
set -e
P=${1:-rNN}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$P
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 20 > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 5 --steps 5 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 5 --steps 5 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/sq -- python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 5 --steps 5 > $O/sq.log 2>&1
cd $R
python tools/pmc_summary.py $O/stats $O/fetch $O/write $O/sq --out gpurun_out/${P}
python tools/trace_breakdown.py $O/stats > gpurun_out/${P}_trace_breakdown_513.txt
python tools/level_timing.py > gpurun_out/${P}_level_timing_513.txt 2>&1
python bench.py > gpurun_out/${P}_bench_n1.json 2> gpurun_out/${P}_bench_n1.err
rm -rf $O/stats $O/fetch $O/write $O/sq
tail -c 1500 gpurun_out/${P}_bench_n1.json
