#!/bin/bash
# Run on the GPU box: MFMA-pipe occupancy of every fused kernel instance of the default bench command (PMC pass, counters only).
#   bash tools/pmc_bench_mfma.sh <tag>   -> gpurun_out/pmc_mfma_<tag>.json (+ table on stdout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r01}
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES --output-format csv -d gpurun_out/pmc_mfma_${tag} -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > gpurun_out/pmc_mfma_${tag}.log 2>&1
python - "$tag" <<'PY'
import csv, glob, json, sys, collections, re
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(f"gpurun_out/pmc_mfma_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fused_" not in k:
            continue
        k = re.sub(r"\(bt::FwdArgs\)|void bt::", "", k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[k] += 1
out = {}
print(f"{'kernel instance':70s} launches  MFMA-busy/duration  VALU(non-MFMA)/MFMA instr")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
    # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines
    busy = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / max(1.0, c["SQ_BUSY_CYCLES"] / 32)
    ratio = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / max(1.0, c["SQ_INSTS_MFMA"])
    out[k] = dict(launches=cnt[k], mfma_busy_frac=round(busy, 4), valu_per_mfma=round(ratio, 3))
    print(f"{k[:70]:70s} {cnt[k]:8d}  {busy:18.3f}  {ratio:10.2f}")
json.dump(out, open(f"gpurun_out/pmc_mfma_{tag}.json", "w"), indent=1)
PY
