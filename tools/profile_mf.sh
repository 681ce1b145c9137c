#!/bin/bash
# rocprofv3 evidence for the nested-dissection route on the C4 / C5 problem size (run on the GPU box from the repo root):
#   tools/profile_mf.sh TAG [CELLS]
#   pass 1: --kernel-trace --stats, default streams (4)               -> profiles/TAG_mf_kernel_stats.csv
#   pass 1b: the same on ONE stream (HOMMX_MF_STREAMS=1: kernel durations do not overlap, their sum is the wall clock)
#                                                                     -> profiles/TAG_mf_1stream_kernel_stats.csv
#   pass 2-4: --pmc SQ set | FETCH_SIZE | WRITE_SIZE (counters only, separate passes, never mixed with tracing; one stream, so that the
#             dispatch order is the launch order and tools/mf_groups.py can attribute dispatches to tree levels)
#   summary: profiles/TAG_mf_pmc_summary.json (per kernel family: time share and TFLOP/s = MFMA flops / time on ONE stream, MFMA-busy
#            fraction, HBM bytes per cell), profiles/TAG_mf_groups.json (per tree level)
set -e
TAG=${1:-r03}
CELLS=${2:-512}
export TMPDIR=/tmp
OUT=gpurun_out/prof_mf_$TAG
rm -rf $OUT && mkdir -p $OUT profiles
CMD="python3 tools/mf_check.py --time-only $CELLS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -o run -- $CMD > $OUT/stats4.log 2>&1
export HOMMX_MF_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/sq -o run -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o run -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o run -- $CMD > $OUT/write.log 2>&1
unset HOMMX_MF_STREAMS
python3 tools/pmc_summary.py $OUT/pmc_all.json $OUT/sq $OUT/fetch $OUT/write > $OUT/pmc_summary.log
python3 tools/mf_groups.py $OUT 1 --cells $CELLS --json profiles/${TAG}_mf_groups.json > $OUT/groups.log 2>&1 || true
cp $OUT/groups.log profiles/${TAG}_mf_groups.txt 2>/dev/null || true
python3 tools/mf_summary.py "$TAG" "$OUT" "$CELLS"
