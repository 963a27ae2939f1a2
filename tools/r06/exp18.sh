#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp18; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/valu_rates tools/probes/valu_rates.hip > $O/build.log 2>&1 && /tmp/valu_rates > $O/valu_rates.txt 2>&1
tail -12 $O/valu_rates.txt
