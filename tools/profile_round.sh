#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repository root):
#   bench line, rocprofv3 kernel-trace stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as
#   MI355X_MICROARCH.md prescribes; never combined with a trace domain), HBM traffic summary.
# Output: gpurun_out/prof/ ; copy what should be judged into profiles/.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
BATCH=${BATCH:-512} # bench.py's default resident batch; the PMC summaries are keyed on it
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done: $(cut -c1-200 "$OUT/bench.json")"
# (bench.py reads profiles/pmc_traffic.json and profiles/pmc_valu.json of the PREVIOUS collection; after copying this
#  run's files into profiles/ the next bench line carries them)
# per-launch tables: one internal stream, so that a "launch" is a whole resident batch
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o r -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs > /dev/null 2> "$OUT/pmc_fetch.err"
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o r -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs > /dev/null 2> "$OUT/pmc_write.err"
echo "pmc write done"
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_valu" -o r -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs > /dev/null 2> "$OUT/pmc_valu.err"
echo "pmc valu done"
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc_fetch" "$OUT/pmc_write" $BATCH "$OUT/pmc_traffic.json"
python3 "$ROOT/tools/pmc_valu.py" "$OUT/pmc_valu" $BATCH "$OUT/pmc_valu.json"
find "$OUT/pmc_valu" -name "*counter_collection.csv" -exec cp {} "$OUT/sq_insts_valu_counter_collection.csv" \;
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/pmc_fetch" -name "*counter_collection.csv" -exec cp {} "$OUT/fetch_size_counter_collection.csv" \;
find "$OUT/pmc_write" -name "*counter_collection.csv" -exec cp {} "$OUT/write_size_counter_collection.csv" \;
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_valu"
ls -la "$OUT"
