#!/bin/bash
# Builds tools/_build/ceres_oracle IF Ceres Solver + Eigen3 are installed on this machine; exits 3 (and builds nothing) otherwise.
# Nothing is stubbed or downloaded: a machine without the libraries simply has no Ceres oracle (tests/test_ceres_oracle.py then skips).
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
CERES_H=""
for d in /usr/include /usr/local/include /opt/conda/include "$HOME/.local/include"; do [ -f "$d/ceres/ceres.h" ] && CERES_H="$d" && break; done
EIGEN=""
for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/conda/include/eigen3 /usr/include; do [ -f "$d/Eigen/Core" ] && EIGEN="$d" && break; done
if [ -z "$CERES_H" ] || [ -z "$EIGEN" ]; then echo "ceres_oracle: Ceres / Eigen3 headers not found (ceres: '${CERES_H}', eigen: '${EIGEN}'): not built"; exit 3; fi
mkdir -p "$HERE/_build"
g++ -O2 -std=c++14 -I"$CERES_H" -I"$EIGEN" -o "$HERE/_build/ceres_oracle" "$HERE/ceres_oracle.cpp" -lceres -lglog -lpthread || { echo "ceres_oracle: build failed"; exit 4; }
echo "built $HERE/_build/ceres_oracle"
