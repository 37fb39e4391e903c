#!/bin/bash
# Builds the library of a committed revision (default HEAD) next to the working tree's as lib/libergodic_amd_prev.so,
# for a same-box A/B through tools/ab_variants.sh "main _prev" (the working tree = main).  Uses a scratch worktree.
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=/tmp/eea_prev_wt
rm -rf "$WT"; git -C "$ROOT" worktree prune
git -C "$ROOT" worktree add -f "$WT" "$REV" -q || exit 1
make -C "$WT/ergodic_exploration_amd/csrc" -j8 2>&1 | grep -E "error" -A3
cp "$WT/ergodic_exploration_amd/lib/libergodic_amd.so" "$ROOT/ergodic_exploration_amd/lib/libergodic_amd_prev.so"
git -C "$ROOT" worktree remove --force "$WT"
ls -la "$ROOT/ergodic_exploration_amd/lib/"
