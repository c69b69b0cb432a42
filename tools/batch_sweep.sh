#!/bin/bash
# SURVEY 8d sweep: resident batch sizes at the KITTI configuration, plus the EuRoC (cfg 1) and 1080p (cfg 4) shapes.
# Output: one bench JSON line per run in gpurun_out/sweep.jsonl
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sweep.jsonl
: > "$OUT"
for B in 1 8 64 256 512 1024; do
  python3 "$ROOT/bench.py" --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs >> "$OUT"
done
python3 "$ROOT/bench.py" --batch 512 --width 752 --height 480 --features 1000 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs >> "$OUT"
python3 "$ROOT/bench.py" --batch 8 --width 1920 --height 1080 --features 2000 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs >> "$OUT"
python3 "$ROOT/bench.py" --config 4 --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs >> "$OUT"
python3 "$ROOT/bench.py" --batch 128 --width 1920 --height 1080 --features 2000 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs >> "$OUT"
python3 - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    c = d["config"]
    cpu = d.get("cpu_baseline")
    print("%4dx%-4d N=%d batch %4d: %9.0f frames/s  %.3f ms/step  stages %s%s" % (c["width"], c["height"], c["n_features"],
          c["frames_per_gpu_per_step"], d["value"], d["ms_per_step"], d["stages_ms"],
          "  CPU oracle %.0f frames/s on %d threads" % (cpu["value"], cpu["cores"]) if cpu else ""))
PY
