cd $GRAFT_REPO_ROOT
for m in "" "--graph" "--graph --parallel-renders"; do
  for np in "" "--no-prune"; do
    python examples/train_synthetic.py --gaussians 200000 --size 512 --iters 1000 --sun-altitude-only --random-camera $m $np 2>/dev/null | tail -1 | sed "s/^/[$m $np] /"
  done
done
