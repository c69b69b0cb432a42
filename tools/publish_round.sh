#!/bin/bash
# Copy what tools/collect_round.sh left under gpurun_out/round/ into profiles/ under this round's names.  usage: publish_round.sh r02
set -e
T=${1:?round tag, e.g. r02}
R=gpurun_out/round
cp $R/final_bench.json profiles/${T}_final_bench.json
cp $R/bench_under_rocprof.json profiles/${T}_final_bench_under_rocprof.json
cp $R/kernel_stats.csv profiles/${T}_final_kernel_stats.csv
cp $R/kernel_stats_overlapped.txt profiles/${T}_kernel_stats_overlapped.txt
cp $R/sq_breakdown.txt profiles/${T}_sq_breakdown.txt
cp $R/lds_breakdown.txt profiles/${T}_lds_breakdown.txt
cp $R/batch_sweep.txt profiles/${T}_batch_sweep.txt
cp $R/single_frame_latency.txt profiles/${T}_single_frame_latency.txt
cp $R/match_latency.txt profiles/${T}_match_latency.txt
cp $R/ba_latency.txt profiles/${T}_ba_latency.txt
cp $R/bow_timing.txt profiles/${T}_bow_timing.txt
cp $R/octree_phases.txt profiles/${T}_octree_phases.txt
cp $R/octree_phases_1080.txt profiles/${T}_octree_phases_1080.txt
cp $R/tcp_counters.txt profiles/${T}_tcp_counters.txt
cp $R/fast_cell_times.txt profiles/${T}_fast_cell_times.txt
cp $R/frame_timeline_1080.txt profiles/${T}_frame_timeline_1080.txt
cp $R/frame_timeline_kitti.txt profiles/${T}_frame_timeline_kitti.txt
cp $R/kernel_stats_match.txt profiles/${T}_kernel_stats_match.txt
cp $R/step_timeline.txt profiles/${T}_step_timeline.txt
cp $R/fast_mix.json profiles/fast_mix.json
cp $R/valu_ops3.txt profiles/${T}_valu_ops3.txt
cp $R/fp4_hamming.txt profiles/${T}_fp4_hamming.txt
cp $R/fast_mix.txt profiles/${T}_fast_mix.txt
cp $R/fast_stage_counts.txt profiles/${T}_fast_stage_counts.txt
mkdir -p profiles/${T}_pmc
cp $R/*_counter_collection.csv profiles/${T}_pmc/
cp gpurun_out/prof/pmc_traffic.json gpurun_out/prof/pmc_valu.json profiles/
