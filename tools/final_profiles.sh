#!/bin/bash
# The round's evidence set in one go (run through gpurun; outputs under gpurun_out/final_<tag>/):
#   tools/final_profiles.sh <tag> [part ...]     parts: nearg ongrid sq sizes emul   (default: all)
set -u
TAG=${1:-r}; shift || true
PARTS=${*:-nearg ongrid sq sizes emul}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
cd $ROOT
for part in $PARTS; do
  echo "== $part"
  case $part in
    nearg) bash tools/profile_round.sh ${TAG}_ng > $OUT/profile_ng.log 2>&1 ;;
    ongrid) bash tools/profile_round.sh ${TAG}_og --method ongrid > $OUT/profile_og.log 2>&1 ;;
    sq) bash tools/pmc_pass.sh ${TAG}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY" \
                                        "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
                                        "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
                                        "TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum" > $OUT/pmc_sq.log 2>&1
        python3 tools/pmc_counters.py gpurun_out/pmc_${TAG}_sq k_brick_masks k_ng_trace k_brick_records k_refine_trace k_edge_flag_listed k_edge_dilate_list > $OUT/pmc_sq_512_neargrid.txt ;;
    sizes) for s in 64 256 1024; do
             python3 bench.py --size $s --steps 5 --warmup 1 --no-cpu --no-dropin --no-config5 --no-user-legs --no-batch > $OUT/bench_$s.json 2> $OUT/bench_$s.err; echo "size $s rc=$?"
           done
           ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_1024 -o ks -- python3 $ROOT/bench.py --size 1024 --steps 3 --warmup 1 --no-cpu --no-dropin --no-config5 --no-user-legs --no-batch --no-sustained --no-stage-steps > $OUT/stats_1024.log 2>&1 ) ;;
    emul) timeout -k 10 400 python3 tools/slab_emulation.py --size 512 --ranks 1 2 4 8 --halo 16 --steps 10 --warmup 3 > $OUT/slab_emulation_512.jsonl 2> $OUT/slab_emulation_512.err; echo "emul 512 rc=$?"
          timeout -k 10 600 python3 tools/slab_emulation.py --size 1024 --ranks 1 2 4 8 --halo 16 --margin 64 --steps 3 --warmup 1 > $OUT/slab_emulation_1024.jsonl 2> $OUT/slab_emulation_1024.err; echo "emul 1024 rc=$?"
          ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_8slabs -o ks -- python3 $ROOT/tools/slab_emulation.py --size 512 --ranks 8 --halo 16 --steps 10 --warmup 3 > $OUT/stats_8slabs.log 2>&1 ) ;;
  esac
done
ls -R $OUT | head -60
