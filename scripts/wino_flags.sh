for f in 0 4 0 4; do echo "== flags $f"; AESR_WINO_FLAGS=$f timeout -k 10 120 python scripts/bench_wino.py ae 2>&1 | grep -E "TOTAL"; done
AESR_WINO_FLAGS=4 timeout -k 10 120 python scripts/bench_wino.py ae --check-only 2>&1 | grep "fwd:" | head -3
