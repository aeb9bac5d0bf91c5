#!/bin/bash
# long runs of the replayed step with the round-4 kernels (one-launch BatchNorm with its grid barrier, one-launch weight preparation, peer
# exchange): a rare race would show as a watchdog count, a NaN or a loss that differs between two identical runs
OUT=$GRAFT_REPO_ROOT/gpurun_out
F=$OUT/r04_soak.txt
: > $F
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1: %d steps, %.3f ms/step, final loss %.6f, ring watchdog %d' % (d['steps'], d['ms_per_step'], d['final_loss'], d['ring_watchdog_timeouts']))" >> $F; }
for RUN in 1 2; do
python3 bench.py --config c2 --triplets 2 --steps 5000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line "c2 2 triplets (one-launch BatchNorm) run $RUN"
done
for RUN in 1 2; do
python3 bench.py --config c3 --triplets 2 --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line "c3 2 triplets (one-launch BatchNorm) run $RUN"
done
python3 bench.py --config c2 --triplets 3 --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line "c2 3 triplets (one-launch BatchNorm, 3 units per workgroup)"
for RUN in 1 2; do
python3 bench.py --config c2 --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line "c2 12 triplets run $RUN"
done
for S in rccl p2p; do for RUN in 1 2; do
AESR_FORCE_DP=1 AESR_SYNCBN=$S python3 bench.py --config c2 --triplets 2 --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line "c2 2 triplets, data parallel (communicator of one, SyncBN exchange $S) run $RUN"
done; done
AESR_BN_FUSED=0 python3 bench.py --config c2 --triplets 2 --steps 3000 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line "c2 2 triplets, AESR_BN_FUSED=0 (three-launch BatchNorm)"
cat $F
