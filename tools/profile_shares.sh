#!/bin/bash
# What ONE rank of an N-GPU run does, on the one GPU of a test box (tools/rank_share.py), for the benchmark world and for
# the same presets on a world with a geography - in ONE gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/profile_shares.sh r04'
# then here:  cp gpurun_out/shares/*.json profiles/      (files are named <tag>_<preset>_<size>_<world>_rank_share_of_<N>[_<rule>].json)
set -euo pipefail
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out/shares
rm -rf "$out"
mkdir -p "$out"
T="timeout -k 10"
run() {   # name, args...
  local name=$1
  shift
  echo "== $name" | tee -a "$out/log.txt"
  $T 420 python3 "$root/tools/rank_share.py" "$@" > "$out/${tag}_$name.json" 2> "$out/$name.log"
  tail -c 400 "$out/${tag}_$name.json" >> "$out/log.txt"
  echo >> "$out/log.txt"
}
# C3 at 10 M agents / 8: the benchmark world (uniformly random) under the per-set rule of rounds 1-3 and under the per-venue
# rule, then the clustered world; / 4 and / 2 on the clustered world
run c3_10m_random_rank_share_of_8_set_rule    --of 8 --exchange-rule set
run c3_10m_random_rank_share_of_8             --of 8
run c3_10m_random_rank_share_of_8_halo_by_id  --of 8 --halo-order id
run c3_10m_clustered_rank_share_of_8          --of 8 --geography clustered --generator torch
run c3_10m_clustered_rank_share_of_4          --of 4 --geography clustered --generator torch
run c3_10m_clustered_rank_share_of_2          --of 2 --geography clustered --generator torch
# C5 at 1e8 agents / 8 (BASELINE config 5 as specified): drawn and cut out on the device
run c5_100m_random_rank_share_of_8            --preset c5 --of 8 --generator torch
run c5_100m_random_rank_share_of_8_halo_by_id --preset c5 --of 8 --generator torch --halo-order id
run c5_100m_clustered_rank_share_of_8         --preset c5 --of 8 --generator torch --geography clustered
# the june preset (the reference's membership structure, eleven networks) / 8
run june_10m_rank_share_of_8                  --preset june --of 8
echo "== done" | tee -a "$out/log.txt"
