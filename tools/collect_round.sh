#!/bin/bash
# Everything profiles/ holds for a round, in one GPU-box call: PMC passes + kernel stats (profile_round.sh), the bench line
# with those PMC files in place, per-kernel breakdowns, the batch sweep and the latency tools.  Output under gpurun_out/round/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/round
mkdir -p $O
cd $ROOT
bash tools/profile_round.sh > $O/profile_round.log 2>&1 || { tail -5 $O/profile_round.log; exit 1; }
cp gpurun_out/prof/pmc_traffic.json gpurun_out/prof/pmc_valu.json profiles/   # the bench line below replays them
echo "pmc done"
# FAST's mix-weighted issue floor: stage counts from the probe build, instruction classes from the ISA (hipcc is on the box)
make -C monoorbslam3_amd/csrc -s -j8 prof > $O/make_prof.log 2>&1 && python3 tools/fast_mix.py 512 > $O/fast_stage_counts.txt 2>&1 \
  && python3 tools/isa_mix.py > $O/fast_mix.txt 2>&1 && cp profiles/fast_mix.json $O/ ; echo "fast mix done"
python3 bench.py > $O/final_bench.json 2> $O/final_bench.err || { tail -5 $O/final_bench.err; exit 1; }
echo "bench done: $(cut -c1-160 $O/final_bench.json)"
bash tools/kernel_stats.sh > $O/kernel_stats_overlapped.txt 2>&1; echo "kernel stats done"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o r -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-density-sweep --no-extra-configs > /dev/null 2>&1; python3 $ROOT/tools/step_timeline.py /tmp/kt2 2 > $O/step_timeline.txt 2>&1 ); echo "timeline done"
bash tools/sq_breakdown.sh > $O/sq.log 2>&1; cp gpurun_out/sq_breakdown.txt $O/; echo "sq done"
bash tools/lds_breakdown.sh > $O/lds.log 2>&1; cp gpurun_out/lds_breakdown.txt $O/; echo "lds done"
bash tools/batch_sweep.sh > $O/batch_sweep.txt 2> $O/batch_sweep.err; echo "sweep done"
{ python3 tools/latency.py 1242 375 2000; python3 tools/latency.py 752 480 1000; python3 tools/latency.py 1920 1080 2000; bash tools/latency_c.sh; } 2>&1 | grep -v amdgpu.ids > $O/single_frame_latency.txt; echo "latency done"
python3 tools/match_latency.py 2>&1 | grep -v amdgpu.ids > $O/match_latency.txt; echo "match latency done"
python3 tools/ba_latency.py 2>&1 | grep -v amdgpu.ids > $O/ba_latency.txt; python3 tools/bow_timing.py 2>&1 | grep -v amdgpu.ids > $O/bow_timing.txt; echo "ba / bow done"
python3 tools/octree_phases.py 1 2>&1 | grep -v amdgpu.ids > $O/octree_phases.txt || true
python3 tools/octree_phases.py 1 1920 1080 2>&1 | grep -v amdgpu.ids > $O/octree_phases_1080.txt || true
{ python3 tools/fast_cell_times.py 1920 1080; python3 tools/fast_cell_times.py 1242 375; python3 tools/fast_cell_times.py 1920 1080 fast_cell_group=1; } 2>&1 | grep -v amdgpu.ids > $O/fast_cell_times.txt || true
bash tools/frame_timeline_host.sh 1920 1080 2000 > $O/frame_timeline_1080.txt 2>&1 || true
bash tools/frame_timeline_host.sh 1242 375 2000 > $O/frame_timeline_kitti.txt 2>&1 || true
bash tools/kernel_stats_match.sh > $O/kernel_stats_match.txt 2>&1 || true; echo "single-frame probes done"
bash tools/ta_breakdown.sh > $O/tcp.log 2>&1; cp gpurun_out/ta_breakdown.txt $O/tcp_counters.txt; echo "tcp done"
for m in valu_ops3 fp4_hamming; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/microbench/$m.hip -o /tmp/$m && /tmp/$m > $O/$m.txt 2>&1; done; echo "microbench done"
cp gpurun_out/prof/*_counter_collection.csv gpurun_out/prof/kernel_stats.csv gpurun_out/prof/bench_under_rocprof.json $O/
ls $O
