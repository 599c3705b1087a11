#!/bin/bash
# same-box A/B/A/B of the captured pretraining step with and without the squeeze-excitation gate taken by the projection on
# load (vision.se_projection).   bash scripts/diag/run_seproj_ab.sh
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "$name: $(env "$@" GRAPH=1 STEPS=10 python3 scripts/diag/time_pretrain_step.py 2>&1 | tail -1)"; }
run apart_a SEPROJ=0 && run fused_a SEPROJ=1 && run apart_b SEPROJ=0 && run fused_b SEPROJ=1
