#!/bin/bash
# stem forward inside the replayed pretraining step for several row-chunk counts (diagnostic library)
cd $GRAFT_REPO_ROOT
for c in 15 30 60 120; do
  IAS_HIP_LIB=$GRAFT_REPO_ROOT/inverse-audio-synthesis_amd/csrc/libias_hip_diag.so IAS_STEM_FWD_CHUNKS=$c bash scripts/diag/trace_pretrain_list.sh > /dev/null 2>&1
  echo "chunks $c: $(grep stem_fwd gpurun_out/trace_ptl/list.txt | cut -c1-30)"
done
