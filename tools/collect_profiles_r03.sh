# Profile collection for the heavy BASELINE configurations on the GPU box:
#   gpurun -- bash tools/collect_profiles_r03.sh [tag]
# raw output under gpurun_out/<tag>/<config>/..., summarised into profiles/r03/ by tools/summarize_r03.py
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM"
SQ4="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES"
SQ3="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES"
run_cfg () {
    name=$1; shift
    D=$O/$name
    mkdir -p $D
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $D/bench_under_rocprof.log 2>&1
    timeout 900 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $D/bench.log 2>&1
    PMCCMD="python $R/bench.py --steps 1 --warmup 0 --seconds 174.8 --files 1 --no-cpu-baseline --no-extras $*"
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o run -- $PMCCMD > $D/fetch.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o run -- $PMCCMD > $D/write.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $D/sq1 -o run -- $PMCCMD > $D/sq1.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $D/sq2 -o run -- $PMCCMD > $D/sq2.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d $D/sq3 -o run -- $PMCCMD > $D/sq3.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $SQ4 --output-format csv -d $D/sq4 -o run -- $PMCCMD > $D/sq4.log 2>&1
    # keep what travels back small: the per-dispatch CSVs only
    find $D -name '*.db' -delete; find $D -name '*agent_info*' -delete
}
run_cfg M  --config M
run_cfg C2 --config C2
run_cfg C3 --config C3
run_cfg C4 --config C4
run_cfg C5 --config C5
du -sh $O
