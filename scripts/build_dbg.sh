#!/bin/bash
# development aid: libspada_dbg.so = the engine with -DSPADA_TASK_DBG=1 (chain / phase counters printed to stderr); use with
# SPADA_LIB_PATH=$PWD/spada_sim_amd/lib/libspada_dbg.so
set -e
cd "$(dirname "$0")/../spada_sim_amd/csrc"
mkdir -p build/dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-parameter -I../../include -I. -DSPADA_TASK_DBG=1 "$@" -c spada_engine.hip -o build/dbg/spada_engine.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/libspada_dbg.so build/spada_host.o build/spada_cycle.o build/dbg/spada_engine.o -lgomp
