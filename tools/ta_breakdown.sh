#!/bin/bash
# L1 / TLB / texture-data counters per kernel (two separate --pmc passes, no trace domains): what the per-key-point
# kernels wait for.  Output: gpurun_out/ta_breakdown.txt
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tab
# (a first pass with GRBM_GUI_ACTIVE + TA_TA_BUSY / TA_ADDR_STALLED_BY_TC / TA_DATA_STALLED_BY_TC hung rocprofv3 on this pool in
# round 2 and was killed after 7 minutes of silence with an empty log -- profiles/README.md; that set is not collected any more)
P2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
P3="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum"
i=0
for P in "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d /tmp/tab/p$i -o r -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs --batch 256 > /dev/null 2> /tmp/tab$i.err || { tail -5 /tmp/tab$i.err; exit 1; }
done
python3 - <<'PY' | tee "$ROOT/gpurun_out/ta_breakdown.txt"
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/tab/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith('k_'):
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
names = sorted({n for c in acc.values() for n in c})
for k, c in sorted(acc.items()):
    print(k)
    for n in names:
        if n in c:
            print("    %-44s %14.4g" % (n, sum(c[n]) / len(c[n])))
PY
