#!/bin/bash
# development aid: scripts/dev/ru.sh [extra -D flags]  -> registers, spills and scratch of every k_task instantiation
# (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel
cd "$(dirname "$0")/../../spada_sim_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-parameter -I../../include -I. -Rpass-analysis=kernel-resource-usage "$@" -c spada_engine.hip -o /tmp/eng_ru.o 2>&1 | grep -A11 "Function Name: .*k_task" | grep "Function Name\|TotalSGPRs\|VGPRs:\|Scratch\|Spill\|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - - - - | sed 's/Function Name: _ZN5spada//; s/NS_8TaskArgsE//'
