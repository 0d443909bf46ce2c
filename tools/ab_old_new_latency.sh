# Dev tool: single-frame latency (tools/lat_default.py) of the tree under _old/ (an older commit, built) and of this tree, alternating, same box
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for t in _old .; do echo "== $t"; (cd $t && timeout -k 10 200 python tools/lat_default.py 2>&1 | grep "cap_o 1 ") || exit 1; done
done
