#!/bin/bash
# one rocprofv3 --pmc pass of the bench workload per argument (a quoted counter list each; at most ~2 TA/TCP/TD or 8 SQ
# counters fit one pass -- a list the hardware cannot collect makes rocprofv3 abort, hence the timeout):
#   tools/pmc_pass.sh <tag> "SQ_WAVE_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for counters in "$@"; do
  timeout -k 10 180 rocprofv3 --pmc $counters --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-dropin --no-odd --no-user-legs --no-batch --no-sustained --no-stage-steps --steps 1 --warmup 0 > $OUT/p$i.log 2>&1
  echo "pass $i ($counters): rc=$?"
  i=$((i+1))
done
ls $OUT
