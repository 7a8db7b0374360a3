#!/bin/bash
# Round 6's ONE closing GPU call (VERDICT r5 hygiene: one re-take of profiles/ per round):
#   tools/profile_round.sh r06   default bench line + kernel stats + PMC passes of the headline + the other modes / configurations
#   + the exact-fp32 (Winograd) step under its own PMC passes: SQ / GRBM (clock, matrix busy) and FETCH_SIZE / WRITE_SIZE
#   + the tensorizer stand-alone
# then:  bash tools/install_profiles.sh r06  and copy gpurun_out/r06/r06_* into profiles/
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1
bash tools/r6_pmc.sh r06/fp32_pmc > /dev/null 2>&1; cp $O/fp32_pmc/pmc_sq.md $O/r06_pmc_sq_fp32_single_stream.md
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/fp32_$c
  EVR_FP32=1 rocprofv3 --kernel-trace --pmc $c -d $O/fp32_$c -o p -- python $R/bench.py --sub --no-overlap --cpu-frames 0 --parity-frames 1 --steps 4 --warmup 2 > /dev/null 2> $O/fp32_$c.err
  python $R/tools/rocpd_pmc.py $(ls $O/fp32_$c/*.db $O/fp32_$c/*/*.db 2>/dev/null | head -1) | head -12 > $O/r06_pmc_${c}_fp32.md
  rm -rf $O/fp32_$c
done
cd $R
python tools/voxel_bench.py --windows 64 512 2048 > $O/voxelizer_standalone.jsonl 2>/dev/null
python tools/voxel_bench.py --windows 64 512 --sensor 640x480 >> $O/voxelizer_standalone.jsonl 2>/dev/null
tail -c 1200 $O/bench_default.json; echo; head -8 $O/r06_pmc_sq_fp32_single_stream.md | cut -c1-160
