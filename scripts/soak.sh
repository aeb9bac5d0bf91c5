#!/bin/bash
# long runs of the replayed step: rare races would show as a watchdog count, a NaN or a loss that differs between two identical runs
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/r03_soak.txt
for CFG in c2 c3; do for RUN in 1 2; do
python3 bench.py --config $CFG --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$CFG run $RUN: 3000 steps, %.3f ms/step, final loss %.6f, ring watchdog %d' % (d['ms_per_step'], d['final_loss'], d['ring_watchdog_timeouts']))" >> $OUT/r03_soak.txt
done; done
python3 bench.py --config c3 --triplets 2 --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3 2 triplets: 3000 steps, %.3f ms/step, final loss %.6f, ring watchdog %d' % (d['ms_per_step'], d['final_loss'], d['ring_watchdog_timeouts']))" >> $OUT/r03_soak.txt
cat $OUT/r03_soak.txt
# one rank of a data-parallel run (communicator of one), both graph forms
for MODE in segments whole; do
AESR_FORCE_DP=1 AESR_DP_GRAPH=$MODE python3 bench.py --config c3 --triplets 2 --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3 2 triplets, data parallel ($MODE, communicator of one): 3000 steps, %.3f ms/step, final loss %.6f, ring watchdog %d' % (d['ms_per_step'], d['final_loss'], d['ring_watchdog_timeouts']))" >> $OUT/r03_soak.txt
done
tail -2 $OUT/r03_soak.txt
