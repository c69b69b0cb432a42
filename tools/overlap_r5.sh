#!/bin/bash
# Round 5, item "make the step overlap": the previous batch's match beside this batch's latency-bound middle
# (quadtree + k_desc_bins + orientation), with the match holding a FIXED share of every CU (ORBM_VAR_BEST2_RESIDENT) so that the
# other kernels always find registers and LDS.  One bench line per setting, then a kernel-trace timeline of the settings named
# in TIMELINES.  Output: gpurun_out/overlap_r5/.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/overlap_r5; mkdir -p $O
COMMON="--no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs --steps 30 --warmup 5"
run() { # name, flags
  python3 bench.py $COMMON $2 > $O/b_$1.log 2>&1 || { echo "$1 FAILED"; tail -3 $O/b_$1.log; return; }
  python3 - "$1" $O/b_$1.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
s = d.get("stages_ms_in_step") or {}
print("%-28s %8.0f frames/s  %.3f ms/step   in step: %s" % (sys.argv[1], d["value"], d["ms_per_step"],
      "  ".join("%s %.3f" % (k, v) for k, v in s.items() if v is not None)), flush=True)
PY
}
run eager_r0 ""
run eager_r1 "--best2-resident 1"
run eager_r2 "--best2-resident 2"
run afterfast_r0 "--match-placement after-fast"
run afterfast_r1 "--match-placement after-fast --best2-resident 1"
run afterfast_r2 "--match-placement after-fast --best2-resident 2"
for extra in "$@"; do run "x_$(echo $extra | tr -c 'a-zA-Z0-9=\n' '_')" "$extra"; done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for t in ${TIMELINES:-eager_r0 afterfast_r1}; do
  case $t in
    eager_r0) F="";; eager_r1) F="--best2-resident 1";; eager_r2) F="--best2-resident 2";;
    afterfast_r0) F="--match-placement after-fast";; afterfast_r1) F="--match-placement after-fast --best2-resident 1";;
    afterfast_r2) F="--match-placement after-fast --best2-resident 2";; *) F="$t";;
  esac
  rm -rf /tmp/tl_$$; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$$ -- python3 $R/bench.py $COMMON --steps 6 --warmup 3 $F > $R/$O/tl_$t.log 2>&1
  python3 $R/tools/step_timeline.py /tmp/tl_$$ 2 > $R/$O/timeline_$t.txt 2>&1
  echo "== timeline $t"; tail -1 $R/$O/timeline_$t.txt
done
