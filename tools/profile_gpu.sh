#!/bin/bash
# rocprofv3 evidence for bench.py, to be run ON THE GPU BOX from the repo root:
#   gpurun -- 'bash tools/profile_gpu.sh r02 replay'        (configs: replay | particles | dense)
# Writes gpurun_out/prof_<tag>_<config>/ (scratch) and the judged summaries into profiles/ via
# tools/summarize_profiles.py (gpurun merges gpurun_out/ only, so the summaries are also left
# under gpurun_out/prof_<tag>_<config>/profiles/ - copy them into profiles/ and commit).
#  1. --kernel-trace --stats of the default bench command of that config (replay: 2 lanes of 32 trajectories,
#     durations include overlap)
#  2. the same for one lane (kernels back to back: their stand-alone durations)
#  3. --pmc passes, ONE counter group per run and nothing else enabled (the pool refuses
#     --pmc combined with other trace domains), one lane, particle batches in one piece: HBM traffic, cache
#     requests, occupancy and waits, and the hardware's own count of float64 / conversion / transcendental
#     instructions (what bench.py prices a kernel's vector issue with).
set -u
TAG=${1:-r02}
CFG=${2:-replay}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
case $CFG in
  replay) STEPS="--steps 20 --warmup 5"; PSTEPS="--steps 3 --warmup 2";;          # (the driver's form; a step is 32 replays of the trajectory)
  particles) STEPS="--steps 12 --warmup 3"; PSTEPS="--steps 6 --warmup 6";;   # (PMC figures from the later half of the launches: settled maps)
  *)      STEPS="--steps 12 --warmup 3"; PSTEPS="--steps 3 --warmup 1";;
esac
COMMON="--config $CFG --no-cpu-baseline --no-single-stream --no-other-configs --sustain-seconds 0 --particle-chunks 1"
# shellcheck disable=SC2086
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 "$R/bench.py" $COMMON $STEPS \
    > "$OUT/bench_default_under_rocprof.json" 2> "$OUT/trace_default.err" || { echo "kernel trace (default) failed"; tail -5 "$OUT/trace_default.err"; exit 1; }
# one lane, every configuration: kernels back to back, i.e. the stand-alone durations the rooflines are priced with
# shellcheck disable=SC2086
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1lane" -- python3 "$R/bench.py" $COMMON $STEPS --lanes 1 \
    > "$OUT/bench_1lane_under_rocprof.json" 2> "$OUT/trace_1lane.err" || { echo "kernel trace (1 lane) failed"; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU"; do
    i=$((i + 1))
    # shellcheck disable=SC2086
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 "$R/bench.py" $COMMON $PSTEPS --lanes 1 --no-parity \
        > /dev/null 2>> "$OUT/pmc.err" || { echo "pmc pass $i failed"; tail -5 "$OUT/pmc.err"; exit 1; }
    echo "$grp" > "$OUT/pmc_$i/counters.txt"
done
if [ "$CFG" = replay ]; then
    # instruction count of k_icp in the launch shape the overlapped default run uses (three queries per lane)
    # shellcheck disable=SC2086
    rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_7" -- python3 "$R/bench.py" $COMMON $PSTEPS --lanes 1 --no-parity --icp-qpt 3 \
        > /dev/null 2>> "$OUT/pmc.err" || { echo "pmc pass 7 failed"; tail -5 "$OUT/pmc.err"; exit 1; }
    echo "SQ_INSTS_VALU icp_qpt=3" > "$OUT/pmc_7/counters.txt"
fi
[ "$CFG" = replay ] && python3 "$R/tools/isa_mix.py" "$OUT/isa_mix.json" > /dev/null
# the summary first (it puts this run's PMC figures under profiles/ of this copy of the tree), then the plain bench line that
# reads them, then the summary again to file that line with the rest
python3 "$R/tools/summarize_profiles.py" "$OUT" "$TAG" "$CFG" > /dev/null
python3 "$R/bench.py" --config $CFG --no-other-configs $STEPS > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
python3 "$R/tools/summarize_profiles.py" "$OUT" "$TAG" "$CFG"
