#!/bin/bash
# development aid: scripts/build_variant.sh NAME [-DFLAG ...]  ->  spada_sim_amd/lib/libspada_NAME.so (release flags plus the
# given ones); use with SPADA_LIB_PATH=$PWD/spada_sim_amd/lib/libspada_NAME.so for A/B measurements
set -e
NAME=$1; shift
cd "$(dirname "$0")/../spada_sim_amd/csrc"
mkdir -p build/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-parameter -I../../include -I. "$@" -c spada_engine.hip -o build/$NAME/spada_engine.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/libspada_$NAME.so build/spada_host.o build/spada_cycle.o build/$NAME/spada_engine.o -lgomp
