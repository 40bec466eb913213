#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the training step (bench.py --train). Output: gpurun_out/prof_<tag>_train_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r03}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_train -- python bench.py --train --steps ${2:-5} --warmup 2 --no-cpu-baseline --no-traffic --no-extras --no-parity > gpurun_out/prof_${tag}_train.log 2>&1
f=$(find gpurun_out/prof_${tag}_train -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/prof_${tag}_train_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms, {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:25]:
    print(f"  {r['Percentage']:>6}%  {int(r['Calls']):5d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:130]}")
PY
tail -2 gpurun_out/prof_${tag}_train.log
