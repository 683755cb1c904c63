#!/bin/bash
# Rank 0's GPU timeline in a two-rank run sharing the box's one GPU (gloo; the halo goes through host memory):
# kernels and memory copies of rank 0 under rocprofv3, rank 1 beside it unprofiled.  Run on the GPU box:
#   bash tools/overlap_trace.sh [tag] [extra bench.py flags]
# then  python3 tools/overlap_timeline.py gpurun_out/ovt_<tag> profiles/<tag>_overlap_timeline.json
tag=${1:-r02}; shift
out=gpurun_out/ovt_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0 SPHX_DIST_BACKEND=gloo SPHX_BENCH_SINGLE=0
args="--gpus 2 --particles 500000 --steps 6 --warmup 4 --no-cpu $*"
RANK=1 timeout -k 10 400 python3 bench.py $args > $out/rank1.out 2>&1 &
r1=$!
RANK=0 timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace -d $out/prof -o r0 --output-format csv -- python3 bench.py $args > $out/rank0.out 2> $out/rank0.err
rc=$?
wait $r1
echo "rank0 exit $rc, rank1 exit $?"
ls $out/prof | head
