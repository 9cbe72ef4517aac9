#!/bin/bash
# round 4: what a firing prune costs the recorded iteration (every prune that removes something re-records the graph), and the
# deferred prune (--defer-prune K: retire now, compact at every K-th prune point) beside it. 200 k Gaussians / 512^2, 1000 iterations.
# PRUNE_EVERY=1 is the reference's cadence (train_pan.py:673-678).
cd $GRAFT_REPO_ROOT
N=${PRUNE_EVERY:-50}; K=${DEFER_K:-10}
for m in "" "--graph" "--graph --parallel-renders"; do
  for np in "--prune-every $N" "--prune-every $N --defer-prune $K" "--no-prune"; do
    python examples/train_synthetic.py --gaussians 200000 --size 512 --iters 1000 --sun-altitude-only --random-camera $m $np 2>/dev/null | tail -2 | tr '\n' ' ' | sed "s/^/[$m $np] /"; echo
  done
done
