# Round-end profile collection on the GPU box (gpurun -- bash tools/collect_profiles.sh); summaries go to profiles/ by hand (profiles/r01/README.md)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fin
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python $R/bench.py --steps 5 --warmup 2 > $O/bench_under_rocprof.log 2>&1
timeout 600 python $R/bench.py > $O/bench.log 2>&1
PMCCMD="python $R/bench.py --steps 1 --warmup 0 --seconds 174.8 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $PMCCMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $PMCCMD > $O/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq1 -o run -- $PMCCMD > $O/sq1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $O/sq2 -o run -- $PMCCMD > $O/sq2.log 2>&1
ls -la $O $O/*
du -sh $O
