run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --repeats 2 --no-split-variant --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('windows_ms'), d.get('frozen_ms_per_step'))"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run AMD_OPT_FLUSH=0
run A=1
