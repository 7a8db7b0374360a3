# PMC passes over the stand-alone tensorizer (512 windows): bash tools/voxel_pmc.sh <tag>   (tables under gpurun_out/voxpmc_<tag>/)
TAG=${1:-x}
R=$PWD; O=$R/gpurun_out/voxpmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o p -- python $R/tools/voxel_bench.py --windows 512 --iters 4 > $O/$n.json 2> $O/$n.err
  python $R/tools/rocpd_pmc.py $(ls $O/$n/*.db $O/$n/*/*.db 2>/dev/null | head -1) > $O/$n.md 2>> $O/$n.err; rm -rf $O/$n; grep "vox_" $O/$n.md; }
pass a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES
pass c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $R
