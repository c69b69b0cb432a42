set -e
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -w $GRAFT_REPO_ROOT/tools/microbench/fetch_calib.hip -o /tmp/fetch_calib
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/calib_out -o r -- /tmp/fetch_calib > $GRAFT_REPO_ROOT/gpurun_out/fetch_calib.txt 2> /tmp/calib.err
python3 - <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/fetch_calib.txt
import csv, glob
BYTES = 16384 * 24576
for f in glob.glob('/tmp/calib_out/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].split('(')[0]
        kb = float(row['Counter_Value'])
        print("%-40s FETCH_SIZE %12.1f KB = %.3f x image bytes" % (name[:40], kb, kb * 1024 / BYTES))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/fetch_calib.txt
