# tile sweep of the Winograd weight gradient on the AE layer table
for t in "8,32" "4,32" "16,8" "8,16" "16,16"; do echo "== tile $t"; AESR_WGRAD_WINO_TILE=$t timeout -k 10 200 python scripts/bench_conv.py ae 2>&1 | sed "s/.*| wgrad/wgrad/" | tail -13 | tr '\n' ' ' | sed 's/wgrad/\n wgrad/g; s/TOTAL/\nTOTAL/g' | grep -v "TOTAL fwd\|TOTAL dgrad"; done
