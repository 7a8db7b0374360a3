R=$PWD; O=$R/gpurun_out/r03f; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/voxel_bench.py --windows 64 512 2048 > $O/voxelizer_standalone.jsonl 2>/dev/null
python tools/voxel_bench.py --windows 64 512 --sensor 640x480 >> $O/voxelizer_standalone.jsonl 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --sub > $O/bench_under_rocprof.json 2> $O/rocprof.err
cd $R
python tools/rocpd_stats.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) --md > $O/kernel_stats.md
rm -rf $O/prof
bash tools/voxel_pmc.sh r03f > $O/voxel_pmc.txt 2>&1
tail -c 600 $O/bench_default.json; grep vox_ $O/kernel_stats.md | cut -c1-100; cat $O/voxelizer_standalone.jsonl | grep '"stats": true' | cut -c1-30,100-200
