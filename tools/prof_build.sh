#!/bin/bash
# rocprofv3 kernel statistics of the structure build (shim incremental timing script): which build kernels cost what
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/buildprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/st
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python $REPO/tools/time_shim_incremental.py > /dev/null 2> $OUT/err.txt
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/build_kernel_stats.txt "time_shim_incremental.py (2 rebuilds + 2 appends at cfg3)" > /dev/null
rm -rf $OUT/st
grep -v "k_schur\|k_pcg\|k_point_\|k_cam_\|k_finalize\|k_lm_\|k_begin\|k_colnorm\|k_xnorm" $OUT/build_kernel_stats.txt | head -40 | cut -c1-110,114-160
