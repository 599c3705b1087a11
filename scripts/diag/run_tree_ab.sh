#!/bin/bash
# Same-box A/B of a baseline tree (a git worktree built under scripts/diag/_bin/basetree: `git worktree add -f
# scripts/diag/_bin/basetree <commit> && make -C scripts/diag/_bin/basetree/inverse-audio-synthesis_amd/csrc`) against the
# current tree, one script, alternating:   bash scripts/diag/run_tree_ab.sh scripts/diag/time_pretrain_step.py GRAPH=1 STEPS=20
R=$GRAFT_REPO_ROOT
script=$1; shift
for rep in 1 2; do
  for tree in base new; do
    if [ $tree = base ]; then cd $R/scripts/diag/_bin/basetree; else cd $R; fi
    echo -n "$tree: "; env "$@" python3 $script 2>/dev/null | tail -1
  done
done
