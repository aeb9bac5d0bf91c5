cd $GRAFT_REPO_ROOT
for cfg in "c2 0" "c5 1"; do set -- $cfg
for pc in default 0 1; do
  if [ $pc = default ]; then unset DEBUG_CLR_GRAPH_PACKET_CAPTURE; else export DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc; fi
  ms=$(python3 bench.py --config $1 $( [ $2 != 0 ] && echo --triplets $2 ) --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$1 triplets=$2 DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc: $ms ms/step"
done; done
