#!/bin/bash
# Per-kernel counter diagnosis (runs ON THE GPU BOX): several --pmc passes of a short bench run, one table out.
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/pmc_diag; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic "$@" > /dev/null 2> $OUT/p$i.err
done
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("sfmba::","")
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
names=sorted(acc, key=lambda k:-acc[k].get("SQ_BUSY_CYCLES",0)/max(cnt[k].get("SQ_BUSY_CYCLES",1),1))
cols=sorted({c for k in acc for c in acc[k]})
with open("$OUT/summary.txt","w") as o:
    for k in names[:12]:
        o.write(k+"\n")
        for c in cols:
            if c in acc[k]: o.write("    %-40s %16.1f\n"%(c, acc[k][c]/cnt[k][c]))
print(open("$OUT/summary.txt").read())
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
