set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
MGX_DELAYS=80 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dist_trace -- python3 $R/tools/dist_exposure.py 257 2 16 > $R/gpurun_out/dist_trace.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
rows=[]
for p in glob.glob('gpurun_out/dist_trace/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Queue_Id'), r.get('Stream_Id')))
rows.sort()
# last 400 kernels: print a window
t0=rows[-400][0]
for s,e,n,q,st in rows[-400:-280]:
    print("%9.1f %9.1f %7.1f q=%s s=%s %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,q,st,n))
PY
