#!/bin/bash
# development aid (this container): builds libspada_abl<mask>.so for the given SPADA_ABLATE masks (WRONG results: stages of the batch task
# left out) -- scripts/dev/ablate_run.sh then times the task kernel of each on the GPU box
cd "$(dirname "$0")/../.."
for m in "$@"; do bash scripts/build_variant.sh abl$m -DSPADA_ABLATE=$m & done
wait
ls -la spada_sim_amd/lib/libspada_abl*.so
