#!/bin/bash
# same-box A/B/A/B of two builds of the product library on the captured pretraining step (scripts/diag/time_pretrain_step.py,
# GRAPH=1): scripts/diag/_bin/libias_hip_base.so (built from the tree one compares against) and the in-tree library.
#   bash scripts/diag/run_pretrain_lib_ab.sh
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/scripts/diag/_bin/libias_hip_base.so
run() { name=$1; shift; echo "$name: $(env "$@" GRAPH=1 STEPS=10 python3 scripts/diag/time_pretrain_step.py 2>&1 | tail -1)"; }
run base_a IAS_HIP_LIB=$B && run new_a A=1 && run base_b IAS_HIP_LIB=$B && run new_b A=1
