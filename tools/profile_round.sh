#!/bin/bash
# The measurements kept under profiles/ for one round, in ONE gpurun call (same box for all of them):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03'
# then, here:  python tools/collect_profiles.py r03
# Every step writes under gpurun_out/final/; a step that fails or times out stops the script (no further GPU step
# after a kill).  Under rocprofv3 the program itself follows `--` (no shell, no env wrapper).
set -euo pipefail
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out/final
rm -rf "$out"
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
T="timeout -k 10"
W="--world-cache /tmp/gj_worlds"

echo "== default bench (the driver's command)" | tee -a "$out/log.txt"
$T 600 python3 "$root/bench.py" > "$out/bench.json" 2> "$out/bench.log"
tail -c 600 "$out/bench.json" >> "$out/log.txt"

echo "== kernel stats" | tee -a "$out/log.txt"
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- \
  python3 "$root/bench.py" --only-headline --steps 30 --warmup 3 --repeats 1 $W > "$out/stats.log" 2>&1

echo "== HBM traffic: FETCH_SIZE, WRITE_SIZE (separate passes)" | tee -a "$out/log.txt"
$T 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o f -- \
  python3 "$root/bench.py" --only-headline --steps 8 --warmup 2 --repeats 1 $W > "$out/pmc_fetch.log" 2>&1
$T 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o w -- \
  python3 "$root/bench.py" --only-headline --steps 8 --warmup 2 --repeats 1 $W > "$out/pmc_write.log" 2>&1

echo "== backward (kept sums; memory-lean form) + its kernel stats" | tee -a "$out/log.txt"
$T 300 python3 "$root/bench.py" --backward --backward-steps 4 --repeats 5 $W > "$out/backward.json" 2> "$out/backward.log"
$T 300 python3 "$root/bench.py" --backward --backward-recompute --backward-steps 4 --repeats 5 $W \
  > "$out/backward_recompute.json" 2> "$out/backward_recompute.log"
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/backward_stats" -o b -- \
  python3 "$root/bench.py" --backward --backward-steps 4 --repeats 5 $W > "$out/backward_stats.log" 2>&1

echo "== C2" | tee -a "$out/log.txt"
$T 300 python3 "$root/bench.py" --preset c2 --no-cpu-baseline > "$out/c2.json" 2> "$out/c2.log"
echo "== done" | tee -a "$out/log.txt"
