#!/bin/bash
# Same-box A/B of this tree against a worktree of an earlier commit (its own library AND its own Python binding):
#   git worktree add .r05ref <commit> && make -C .r05ref/web-splat_amd     then     bash scripts/ab_tree.sh .r05ref "<cmd relative to a tree>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REF=$1; shift
for rep in 1 2; do
  for t in . $REF; do
    echo "== tree $t rep $rep: $(cd $t && eval "$@" 2>&1 | grep -E "frames/s|fps" | tr '\n' ' ' | cut -c1-400)"
  done
done
