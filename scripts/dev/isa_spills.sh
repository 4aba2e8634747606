#!/bin/bash
# development aid: scripts/dev/isa_spills.sh <mode 0|1|2> [extra -D flags]  -> scratch stores / loads of k_task<mode> per stage of the
# batch task (between the BT_MARK comments of the three instantiations: hashed, dense, spilled dense), and the static instruction counts
M=$1; shift
cd "$(dirname "$0")/../../spada_sim_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-parameter -I../../include -I. "$@" --cuda-device-only -S spada_engine.hip -o /tmp/eng_isa.s 2>/dev/null
K=_ZN5spada6k_taskILi${M}ELi2048EEEvPKNS_8TaskArgsE
awk "/^$K:/,/s_endpgm/" /tmp/eng_isa.s > /tmp/k_isa.s
wc -l /tmp/k_isa.s
grep -n "BT_MARK\|scratch_" /tmp/k_isa.s | awk '{print $1, $2, $3, $4}' | awk 'BEGIN{m="pre"} /BT_MARK/{print m, "stores", n_st+0, "loads", n_ld+0; m=$0; n_st=0; n_ld=0; next} /scratch_store/{n_st++} /scratch_load/{n_ld++} END{print m, "stores", n_st+0, "loads", n_ld+0}'
python3 ../../scripts/dev/isa_stages.py /tmp/eng_isa.s $K
