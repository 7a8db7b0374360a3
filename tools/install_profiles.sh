#!/bin/bash
# copy the summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/ under their round names:  bash tools/install_profiles.sh r05
T=${1:-r05}; O=gpurun_out/$T
cp $O/bench_default.json profiles/${T}_bench_default.json; cp $O/bench_default_full.json profiles/${T}_bench_default_full.json
cp $O/bench_under_rocprof.json profiles/${T}_bench_under_rocprof_nseq64.json; cp $O/kernel_stats.md profiles/${T}_kernel_stats_nseq64.md
for m in mx8 mx6 fp32 hyper color; do cp $O/kernel_stats_$m.md profiles/${T}_kernel_stats_$m.md; done
cp $O/kernel_stats_640.md profiles/${T}_kernel_stats_640x480.md; cp $O/kernel_stats_fire.md profiles/${T}_kernel_stats_firenet.md
cp $O/pmc_fetch_size.md profiles/${T}_pmc_fetch_size.md; cp $O/pmc_write_size.md profiles/${T}_pmc_write_size.md; cp $O/pmc_sq.md profiles/${T}_pmc_sq_nseq64.md
[ -f $O/voxelizer_standalone.jsonl ] && cp $O/voxelizer_standalone.jsonl profiles/${T}_voxelizer_standalone.jsonl
cp $O/pmc_traffic.json profiles/pmc_traffic.json
