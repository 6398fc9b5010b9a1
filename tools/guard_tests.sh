#!/bin/bash
# Run the GPU test-suite against the LDS-guard build of the library (on the GPU box):
#   gpurun -- 'bash tools/guard_tests.sh'
# A kernel that writes past its LDS regions makes the next slam_check_status / host-pointer
# call fail with "wrote past its LDS regions".
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
PKG="$R/a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd"
make -s -C "$PKG/csrc" guard            # always: the guard library must match the sources
SLAM_HIP_LIB="$PKG/libslamhip_guard.so" python -m pytest "$R/tests" -m gpu -x -q "$@"
SLAM_HIP_LIB="$PKG/libslamhip_guard.so" python "$R/bench.py" --no-cpu-baseline --check --steps 8 --warmup 2 | grep -o '"parity.*'
