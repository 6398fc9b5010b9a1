#!/bin/bash
# rocprofv3 evidence for bench.py, to be run ON THE GPU BOX from the repo root:
#   gpurun -- 'bash tools/profile_gpu.sh r01'
# Writes gpurun_out/prof_<tag>/ (scratch) and the judged summaries into profiles/ via
# tools/summarize_profiles.py (copy them back: gpurun merges gpurun_out/ only, so the
# summaries are also left under gpurun_out/prof_<tag>/profiles/).
#  1. --kernel-trace --stats of the default bench command (4 lanes: durations include overlap)
#  2. the same for one lane (kernels back to back: their stand-alone durations)
#  3. --pmc passes, ONE counter group per run and nothing else enabled (the pool refuses
#     --pmc combined with other trace domains), one lane.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline \
    > "$OUT/bench_default_under_rocprof.json" 2> "$OUT/trace_default.err" || { echo "kernel trace (default) failed"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1lane" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --lanes 1 \
    > "$OUT/bench_1lane_under_rocprof.json" 2> "$OUT/trace_1lane.err" || { echo "kernel trace (1 lane) failed"; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --lanes 1 \
        > /dev/null 2>> "$OUT/pmc.err" || { echo "pmc pass $i failed"; exit 1; }
    echo "$grp" > "$OUT/pmc_$i/counters.txt"
done
python3 "$R/bench.py" --steps 100 --warmup 10 --check > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; exit 1; }
python3 "$R/tools/summarize_profiles.py" "$OUT" "$TAG"
