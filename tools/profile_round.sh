#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + two PMC passes of the default bench workload.
# Output goes to gpurun_out/prof_<tag>/; tools/pmc_summary.py turns the PMC csv into the text summary.
#   usage: tools/profile_round.sh <tag> [extra bench args]
set -u
TAG=${1:-r}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ks -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-dropin --no-odd --no-user-legs --no-batch --no-sustained --no-stage-steps --steps 5 --warmup 1 "$@" > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-dropin --no-odd --no-user-legs --no-batch --no-sustained --no-stage-steps --steps 1 --warmup 0 "$@" > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-dropin --no-odd --no-user-legs --no-batch --no-sustained --no-stage-steps --steps 1 --warmup 0 "$@" > $OUT/write.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_fetch_write.txt
# the same two counters in steady state: 1 warm-up + 3 steps, the first step dropped (the one-step passes above see
# cold caches and first-touch faults)
mkdir -p $OUT/steady
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/steady/fetch -o pmc -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-dropin --no-odd --no-user-legs --no-batch --no-sustained --no-stage-steps --steps 3 --warmup 1 "$@" > $OUT/steady/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/steady/write -o pmc -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-dropin --no-odd --no-user-legs --no-batch --no-sustained --no-stage-steps --steps 3 --warmup 1 "$@" > $OUT/steady/write.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/steady --steady 4 > $OUT/pmc_fetch_write_steady.txt
python3 $ROOT/bench.py --steps 5 --warmup 1 --no-user-legs --no-batch "$@" > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -40
