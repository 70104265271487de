#!/bin/bash
# L1/L2 request counts per pixel, march vs tile kernels (one PMC pass).  bash tools/gpu_l2.sh <tag>
TAG=${1:-l2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/l2_${TAG}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/tools/prof_driver.py l2 > $OUT/run.log 2>&1
cd $OUT; python - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob('p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        if 'k_chain' not in name and 'k_lut3d' not in name: continue
        rows.setdefault((int(r['Dispatch_Id']), name[:70]), {})[r['Counter_Name']] = float(r['Counter_Value'])
PX = 16 * 2160 * 3840
for (d, n), c in sorted(rows.items()):
    print(d, n, {k: round(v / PX, 3) for k, v in c.items()})
PY
