#!/bin/bash
# Two ranks REHEARSED on one device (gloo control + data plane for the gradients, peer exchange through IPC for the SyncBN sums, 64 workgroups
# per one-launch BatchNorm kernel so that both ranks' kernels are resident together): 2 000 replayed steps, twice, against the all-reduce form
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_p2p_soak.txt
: > $OUT
P=29600
for S in p2p p2p rccl; do
  P=$((P+1))
  AESR_SINGLE_DEVICE=1 AESR_DIST_BACKEND=gloo AESR_SYNCBN=$S AESR_BN_FUSED_NB=64 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 2000 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks on one device, SyncBN exchange $S: %d steps, %.3f ms/step, final loss %.6f, ring watchdog %d' % (d['steps'], d['ms_per_step'], d['final_loss'], d['ring_watchdog_timeouts']))" >> $OUT
done
cat $OUT
