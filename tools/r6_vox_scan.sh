# Round 6 (GPU box): what a fused scan would cost the tensorizer's range kernel (EVR_VOX_SCANPROBE, voxelize.hip) -- standalone calls
# and per-kernel times.   bash tools/r6_vox_scan.sh  -> gpurun_out/r06_vox_scan/
R=$PWD; O=$R/gpurun_out/r06_vox_scan; mkdir -p $O
for p in 0 1 2; do
  echo "== EVR_VOX_SCANPROBE=$p" | tee -a $O/standalone.txt
  EVR_VOX_SCANPROBE=$p python tools/voxel_bench.py --windows 64 512 2>/dev/null | grep '"stats": true' | tee -a $O/standalone.txt
done
cd /tmp && export TMPDIR=/tmp
for p in 0 1 2; do
  rm -rf $O/prof_$p
  EVR_VOX_SCANPROBE=$p rocprofv3 --kernel-trace --stats -d $O/prof_$p -o k -- python $R/tools/voxel_bench.py --windows 512 --iters 30 > /dev/null 2> $O/rocprof_$p.err
  python $R/tools/rocpd_stats.py $(ls $O/prof_$p/*.db $O/prof_$p/*/*.db 2>/dev/null | head -1) --md 2>/dev/null | grep "vox_" | cut -c1-120 | sed "s/^/probe $p: /" | tee -a $O/kernels.txt
  rm -rf $O/prof_$p
done
