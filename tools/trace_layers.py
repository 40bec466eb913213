#!/usr/bin/env python3
"""Per-layer kernel durations of a bench workload from a rocprofv3 kernel trace (graph replay: no host gaps in the numbers).

On the GPU box:  python tools/trace_layers.py <tag> [workload] [steps]   (env such as BT_NO_SKINNY=1 is passed through)
Starts `rocprofv3 --kernel-trace -- python bench.py ...` as a child BEFORE touching the GPU itself, reads the trace CSV, keeps the
fused_* launches in order, folds them by the number of fused launches per step and prints the median duration per position."""
import csv, glob, json, os, statistics, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "t"
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", f"trace_{tag}")
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--steps", str(steps),
       "--warmup", "2", "--no-cpu-baseline", "--no-traffic", "--no-extras", "--no-parity", "--no-roofline"]
log = open(os.path.join(root, "gpurun_out", f"trace_{tag}.log"), "w")
rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=600).returncode
files = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
if rc or not files:
    sys.exit(f"trace failed rc={rc}; see gpurun_out/trace_{tag}.log")
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "fused_" in r["Kernel_Name"] and "bt::" in r["Kernel_Name"]]
# period = the step's number of fused launches: the shortest period p >= 2 for which the name sequence of the last 4 p launches repeats
names = [n for n, _ in fused]
per = None
for p in range(2, 200):
    tail = names[-4 * p:]
    if len(tail) == 4 * p and all(tail[i] == tail[i + p] for i in range(3 * p)):
        per = p
        break
if per is None:
    sys.exit("no period found in the fused launch sequence")
use = fused[-steps * per:] if len(fused) >= steps * per else fused[-(len(fused) // per) * per:]
res = []
for i in range(per):
    d = [use[j][1] for j in range(i, len(use), per)]
    res.append(dict(pos=i, kernel=use[i][0][:90], median_us=statistics.median(d) / 1e3, min_us=min(d) / 1e3, n=len(d)))
tot = sum(r["median_us"] for r in res)
for r in res:
    print(f"{r['pos']:3d} {r['median_us']:8.1f} us (min {r['min_us']:7.1f}) {r['kernel']}")
print(f"sum of medians {tot:.1f} us over {per} fused launches per step")
json.dump(dict(tag=tag, workload=wl, per_step=per, sum_us=tot, launches=res, env={k: v for k, v in os.environ.items() if k.startswith("BT_")}),
          open(os.path.join(root, "gpurun_out", f"trace_{tag}.json"), "w"), indent=1)
