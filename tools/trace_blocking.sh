#!/bin/bash
# kernel timeline of the last of a few blocking BN254 MSMs of 2^$1 pairs (rocprofv3 --kernel-trace + tools/trace_timeline.py) and
# the wall time of a loop of such calls without the profiler.  usage (on the GPU box): tools/trace_blocking.sh 17
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=${1:-20}
python3 $R/tools/blocking_loop.py $L 200 2>/dev/null | grep blocking
rm -rf $R/gpurun_out/blk_trace
rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/blk_trace -- python3 $R/tools/trace_msm.py $((1 << L)) bn254 > $R/gpurun_out/blk_trace.log 2>&1
F=$(find $R/gpurun_out/blk_trace -name '*kernel_trace.csv' | head -1)
python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/blk_timeline_$L.txt
cat $R/gpurun_out/blk_timeline_$L.txt
