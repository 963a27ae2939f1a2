# Profile collection for the heavy BASELINE configurations on the GPU box:
#   gpurun -- bash tools/collect_profiles_r02.sh [tag]
# raw output under gpurun_out/<tag>/<config>/..., summarised into profiles/r02/ by tools/summarize_r02.py
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM"
SQ3="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES"
run_cfg () {
    name=$1; shift
    D=$O/$name
    mkdir -p $D
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o run -- python $R/bench.py --steps 3 --warmup 1 --seconds 300 --no-cpu-baseline "$@" > $D/bench_under_rocprof.log 2>&1
    PMCCMD="python $R/bench.py --steps 1 --warmup 0 --seconds 174.8 --no-cpu-baseline $*"
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o run -- $PMCCMD > $D/fetch.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o run -- $PMCCMD > $D/write.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $D/sq1 -o run -- $PMCCMD > $D/sq1.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $D/sq2 -o run -- $PMCCMD > $D/sq2.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d $D/sq3 -o run -- $PMCCMD > $D/sq3.log 2>&1
    # keep what travels back small: the per-dispatch CSVs only
    find $D -name '*.db' -delete; find $D -name '*agent_info*' -delete
}
run_cfg C3_m4_B4096_V2      --preset 4 --block 4096 --divisions 2
run_cfg C5_m4_B4096_V2_P3   --preset 4 --block 4096 --divisions 2 --ltp 3
run_cfg C4_m4_B8192_V2_P3   --preset 4 --block 8192 --divisions 2 --ltp 3
run_cfg C2m_m4_B4096_V1     --preset 4 --block 4096 --divisions 1
du -sh $O
