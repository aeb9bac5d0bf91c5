for f in 0 4 6; do echo "== flags $f"; AESR_WINO_FLAGS=$f AESR_WINO_DBG=1 timeout -k 10 120 python scripts/bench_wino.py ae --check-only 2>&1 | grep stamps | sed 's/.*|| //' | sed -n '1,2p'; done
for f in 0 4 6; do echo "== flags $f"; AESR_WINO_FLAGS=$f timeout -k 10 120 python scripts/bench_wino.py ae 2>&1 | grep -E "TOTAL|fwd: wino-vs-igemm [^0-9]"; done
