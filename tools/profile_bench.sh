#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats + HBM traffic counters of the default bench command.
# Outputs under gpurun_out/prof_<tag>/ ; summaries are copied to profiles/ by hand (tracked).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r01}
wl=${2:-cfg3}
steps=${3:-10}
args="--workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-traffic --no-extras --no-parity"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- python bench.py $args > gpurun_out/prof_${tag}_stats.log 2>&1
# PMC passes are separate runs with nothing but the counters (FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic --no-extras --no-parity > gpurun_out/prof_${tag}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic --no-extras --no-parity > gpurun_out/prof_${tag}_write.log 2>&1
python - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {}
f = glob.glob(f"gpurun_out/prof_{tag}_stats/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    fused = [r for r in rows if "fused_" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    ft = sum(float(r["TotalDurationNs"]) for r in fused); fc = sum(int(r["Calls"]) for r in fused)
    out["kernel_stats"] = dict(fused_calls=fc, fused_total_ms=ft / 1e6, fused_avg_us=ft / fc / 1e3, fused_share_of_gpu_time=ft / tot)
    print("top kernels:")
    import shutil; shutil.copy(f[0], f"gpurun_out/prof_{tag}_kernel_stats.csv")
    for r in rows[:16]:
        print(f"  {r['Percentage']:>6}%  {int(r['Calls']):5d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:150]}")
for name in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/prof_{tag}_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = "fused" if "fused_" in r["Kernel_Name"] else "other"
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    out[name] = {k: dict(sum=v[0], launches=v[1], per_launch=v[0] / max(1, v[1])) for k, v in agg.items()}
json.dump(out, open(f"gpurun_out/prof_{tag}_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
