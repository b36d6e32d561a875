#!/bin/bash
# The chase of edge_check, round by round (DESIGN.md 4.2 "the front is shared"): a diagnostic build of the library that times every
# round of every workgroup of k_ec_chase by class (queue length, shedding, the long form), run on config 5 (512^3, ongrid + neargrid
# refinement) with and without sharing.  Run through gpurun:  bash tools/ec_probe.sh   -> gpurun_out/ec_probe.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value \
    -DXB_EC_PROBE -DXB_DEBUG_COUNT -o pybader_amd/libbader_hip_ecprobe.so pybader_amd/csrc/bader_hip.hip 2> /dev/null || exit 1
for share in 1 0; do
  echo "== sharing $share"
  XB_LIBRARY=$ROOT/pybader_amd/libbader_hip_ecprobe.so XB_OPT_2=$((share ? 0 : 32)) XB_OPT_DBG=4 timeout -k 10 120 python3 bench.py --method ongrid --steps 1 --warmup 1 \
      --no-cpu --no-dropin --no-odd --no-user-legs --no-batch --no-config5 2>&1 > /dev/null | grep "edge_check sharing\|  rounds" | tail -6
done | tee gpurun_out/ec_probe.txt
