#!/bin/bash
# round 4: what a firing prune costs the recorded iteration (every prune that removes something re-records the graph), and the
# deferred prune (--defer-prune K: retire now, compact at every K-th prune point) beside it. 200 k Gaussians / 512^2, 1000 iterations.
cd $GRAFT_REPO_ROOT
for m in "" "--graph" "--graph --parallel-renders"; do
  for np in "" "--defer-prune 10" "--no-prune"; do
    python examples/train_synthetic.py --gaussians 200000 --size 512 --iters 1000 --sun-altitude-only --random-camera $m $np 2>/dev/null | tail -2 | tr '\n' ' ' | sed "s/^/[$m $np] /"; echo
  done
done
