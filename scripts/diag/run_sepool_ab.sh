#!/bin/bash
# same-box A/B/A/B of the captured pretraining step with and without the squeeze-excitation pool taken by the normalisation in
# front of the block (BatchNormAct2d.forward(pool=True)).   bash scripts/diag/run_sepool_ab.sh
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "$name: $(env "$@" GRAPH=1 STEPS=10 python3 scripts/diag/time_pretrain_step.py 2>&1 | tail -1)"; }
run own_a SEPOOL=0 && run norm_a SEPOOL=1 && run own_b SEPOOL=0 && run norm_b SEPOOL=1
