#!/bin/bash
# GPU box: rocprofv3 passes over the default bench command (each pass its own run, as MI355X_MICROARCH.md prescribes:
# kernel trace + stats; then --pmc FETCH_SIZE; then --pmc WRITE_SIZE; optionally SQ counters).  Summaries land in
# gpurun_out/prof_<tag>/ ; tools/profile_collect.py turns them into the CSV/JSON files committed under profiles/.
#   usage: bash tools/profile_bench.sh <tag> [sq]
set -e
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 0 --cpu-baseline 0 --partial-pass 0 --bracket-probes 0"
timeout -k 10 560 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.err
echo stats done > $OUT/progress.txt
timeout -k 10 560 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/bench_pmc_fetch.err
echo fetch done >> $OUT/progress.txt
timeout -k 10 560 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- python3 bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/bench_pmc_write.err
echo write done >> $OUT/progress.txt
if [ "$2" = "sq" ]; then
  timeout -k 10 560 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o run -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/bench_pmc_sq.err
  echo sq done >> $OUT/progress.txt
fi
python3 tools/profile_collect.py $OUT $TAG
ls $OUT
