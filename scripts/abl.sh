for a in 0 1 2 3; do echo "=== ablate $a"; SPADA_ABLATE=$a bash scripts/prof_serial.sh a$a webbase 2>&1 | grep -E "k_num_hash|device ms"; done
