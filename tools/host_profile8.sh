#!/bin/bash
# eight ranks' worth of Python on ONE host.  Eight concurrent eager training loops (one process each, pinned to
# disjoint 16-core sets of the box's 128 hardware threads, all sharing the one GPU, small batch so the GPU is not the bound):
# the host enqueue time per step of each, against a single process's.   bash tools/host_profile8.sh > gpurun_out/host_profile8.txt
cd $GRAFT_REPO_ROOT
export HOST_PROFILE_SHORT=1 B=${B:-4}
echo "== one process (cores 0-15)"
taskset -c 0-15 python tools/host_profile.py 2>/dev/null | head -3
echo "== eight processes, cores 16k .. 16k+15"
for k in 0 1 2 3 4 5 6 7; do
  lo=$((16 * k)); hi=$((16 * k + 15))
  ( taskset -c $lo-$hi python tools/host_profile.py 2>/dev/null | head -3 | sed "s/^/[proc $k] /" ) &
done
wait
nproc
