#!/bin/bash
# usage: bash tools/pmc.sh <tag> <microbench args...>   -> gpurun_out/pmc_<tag>_{1,2}/ + summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
P3="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_INSTS_SMEM"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_${tag}_$i -- python tools/microbench.py "$@" --iters 3 > gpurun_out/pmc_${tag}_$i.log 2>&1
  i=$((i+1))
done
python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused_" not in r["Kernel_Name"]:
            continue
        k = r["Counter_Name"]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k in sorted(agg):
    print(f"{k:32s} {agg[k][0]/agg[k][1]:16.0f}  (per launch, {agg[k][1]} launches)")
PY
