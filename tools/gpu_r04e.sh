#!/bin/bash
# round 4, session e: are two fused workgroups resident per CU, and do their phases overlap?  Counters (SQ set 1, 2, 4 + fetch / write /
# tcc) of the fused launch at 4096 clients; the same launch with the odd workgroup slots started late.
TAG=${1:-r04e}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/sdr-server_amd/build/variants
for v in f_stag2 f_stag4; do
  echo "== variant $v"
  XL_LIBRARY_PATH=$V/lib$v.so timeout 120 python tools/group_sweep.py --clients 4096,2048 --groups 8 --poly3 --blocks 160 --opt mix_kernel=2 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_$v.txt
done
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 4096 --groups 8 --modes optimized --blocks 48 --opt mix_kernel=2"
run() { n=$1; shift; timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o p -- $CMD > $OUT/$n.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run sq4 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
per = collections.defaultdict(dict)
for n in ("sq1", "sq2", "sq4", "fetch", "write", "tcc"):
    for f in glob.glob(f"{out}/{n}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("xlp_") and "tables" not in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for extra in ("LDS_Block_Size", "Arch_VGPR_Count", "Accum_VGPR_Count", "Workgroup_Size", "Grid_Size", "Scratch_Size"):
                    if extra in r: per[k][extra] = r[extra]
        for k, d in agg.items():
            for c, v in d.items():
                v = v[3:] if len(v) > 6 else v
                per[k][c] = round(sum(v) / len(v), 1)
    for f in glob.glob(f"{out}/{n}/**/*kernel_trace.csv", recursive=True)[:1]:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("xlp_"): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            v = v[3:] if len(v) > 6 else v
            per[k]["duration_us_" + n] = round(sum(v) / len(v), 1)
for k, d in per.items():
    print(k)
    for c, v in sorted(d.items()): print("    %-28s %s" % (c, v))
PY
find $OUT -name "*.csv" -delete
