#!/bin/bash
# rocprofv3 passes for one workload, each in its own run (gpurun refuses --pmc together with trace domains):
#   tools/profile.sh <tag> <program...>      e.g.  tools/profile.sh r02 python3 bench.py --steps 2 --warmup 1 --cpu-sample 0
# The rocpd databases (tens of MB each) stay on the GPU box: the last step condenses them into profiles/<tag>_summary.json and
# profiles/<tag>_kernel_stats.csv, copied to gpurun_out/ so that they travel back (gpurun merges at most 64 MiB).
# The program itself follows `--` (no env/bash hop: the profiler's preloaded library has already initialised the GPU).
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_stats -- "$@" > $out/prof_${tag}_stats.log 2>&1
echo "stats done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES -d $out/prof_${tag}_sq1 -- "$@" > $out/prof_${tag}_sq1.log 2>&1
echo "sq1 done"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 -d $out/prof_${tag}_sq2 -- "$@" > $out/prof_${tag}_sq2.log 2>&1
echo "sq2 done"
rocprofv3 --pmc FETCH_SIZE -d $out/prof_${tag}_fetch -- "$@" > $out/prof_${tag}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/prof_${tag}_write -- "$@" > $out/prof_${tag}_write.log 2>&1
echo "write done"
python3 tools/summarize_profiles.py $tag "$*" > $out/prof_${tag}_summary.log 2>&1
cp profiles/${tag}_summary.json profiles/${tag}_kernel_stats.csv $out/
rm -rf $out/prof_${tag}_stats $out/prof_${tag}_sq1 $out/prof_${tag}_sq2 $out/prof_${tag}_fetch $out/prof_${tag}_write
echo "summary done"
