#!/bin/bash
# SQ counters of the big kernels of one bench run (two rocprofv3 --pmc passes of eight counters each; run on the GPU
# box):   tools/sq_counters.sh      -> per kernel: wave cycles, VALU-busy, waits, LDS cycles and bank conflicts, ...
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d gpurun_out/pmcA --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM -d gpurun_out/pmcB --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmcB.log 2>&1
python - <<'P'
import csv,glob,collections
for d in ("pmcA","pmcB"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][-60:]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
        for k in acc:
            if "lin_grp" in k or "schur_grp" in k or "backsub" in k:
                print(k, {c: round(v/max(1,n[(k,c)])) for c,v in acc[k].items()})
P
rm -rf gpurun_out/pmcA gpurun_out/pmcB
