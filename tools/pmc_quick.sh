# quick counter comparison of kernel variants: gpurun -- bash tools/pmc_quick.sh <tag> <config> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CFG=$2; shift 2
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/$TAG
mkdir -p $O
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
SQ3="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU"
PMCCMD="python $R/bench.py --steps 1 --warmup 0 --calls-per-step 1 --seconds 174.8 --files 1 --no-cpu-baseline --config $CFG"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/sq1 -o run -- $PMCCMD > $O/sq1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d $O/sq3 -o run -- $PMCCMD > $O/sq3.log 2>&1
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
