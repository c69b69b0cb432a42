#!/bin/bash
# Per kernel: LDS instructions, LDS-array busy cycles and bank-conflict cycles next to the VALU instruction count
# (MI355X_MICROARCH.md, SQ counters; one --pmc pass, no trace domains).  Output: gpurun_out/lds_breakdown.txt
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
BATCH=${BATCH:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ldsb
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ldsb -o r -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs --batch $BATCH > /dev/null 2> /tmp/ldsb.err
python3 - <<'PY' | tee "$ROOT/gpurun_out/lds_breakdown.txt"
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/ldsb/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith('k_'):
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
print("%-20s %10s %12s %12s %12s %14s %14s %10s" % ("kernel", "waves", "VALU/wave", "LDS/wave", "SALU/wave", "LDSact/LDSinst", "conflict/inst", "launches"))
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = m.get('SQ_WAVES', 1) or 1
    li = m.get('SQ_INSTS_LDS', 0) or 1
    print("%-20s %10.3g %12.1f %12.1f %12.1f %14.2f %14.2f %10d" % (k, w, m.get('SQ_INSTS_VALU', 0) / w, m.get('SQ_INSTS_LDS', 0) / w,
          m.get('SQ_INSTS_SALU', 0) / w, m.get('SQ_LDS_IDX_ACTIVE', 0) / li, m.get('SQ_LDS_BANK_CONFLICT', 0) / li, len(c.get('SQ_WAVES', []))))
PY
