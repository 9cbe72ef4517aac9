cd $GRAFT_REPO_ROOT
A="--gaussians 1048576 --size 1024 --iters 24 --no-prune --sun-altitude-only --random-camera"
for x in "" "--parallel-renders" "--graph" "--graph --parallel-renders" "--graph" "--graph --parallel-renders"; do echo "[$x] $(python examples/train_synthetic.py $A $x 2>&1 | tail -1)"; done
python examples/train_synthetic.py --gaussians 30000 --size 192 --iters 100 --sun-altitude-only --random-camera --quiet; python - <<'PY'
import sys; sys.path.insert(0,'examples'); import train_synthetic as t
a=["--gaussians","30000","--size","192","--iters","100","--quiet","--sun-altitude-only","--random-camera"]
print("serial", t.main(a)); print("parallel", t.main(a+["--parallel-renders"])); print("graph parallel", t.main(a+["--parallel-renders","--graph"]))
PY
