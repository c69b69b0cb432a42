#!/bin/bash
# Where the wave cycles of every kernel go (MI355X_MICROARCH.md, SQ counters): parked at s_waitcnt / barriers (WAIT_ANY),
# stalled at issue (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY), and the VALU / VMEM / LDS / SALU shares of the issue cycles.
# SQ counters only, one --pmc pass, no trace domains.  Output: gpurun_out/sq_breakdown.txt
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqb
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/sqb -o r -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs --batch 256 > /dev/null 2> /tmp/sqb.err
python3 - <<'PY' | tee "$ROOT/gpurun_out/sq_breakdown.txt"
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/sqb/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith('k_'):
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
print("%-20s %10s | %% of wave cycles: %8s %10s %8s | %% of wave cycles issuing: %6s %6s %6s %6s" %
      ("kernel", "wave-cyc", "parked", "iss-stall", "issuing", "VALU", "VMEM", "LDS", "SALU"))
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1) or 1
    pct = lambda n: 100.0 * m.get(n, 0) / wc
    print("%-20s %10.3g | %27.1f %10.1f %8.1f | %32.1f %6.1f %6.1f %6.1f" % (k, wc, pct('SQ_WAIT_ANY'), pct('SQ_WAIT_INST_ANY'),
          pct('SQ_ACTIVE_INST_ANY'), pct('SQ_ACTIVE_INST_VALU'), pct('SQ_ACTIVE_INST_VMEM'), pct('SQ_ACTIVE_INST_LDS'), pct('SQ_ACTIVE_INST_SCA')))
PY
