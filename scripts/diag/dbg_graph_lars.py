import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pretrain
SMALL = ["vicreg=fast", "dim=64", "embeddim=256", "vicreg.batch_size=4", "vicreg.mlp=128-128-%d",
         "audio_to_params.batch_size=4", "trainer.log_every=1", "vicreg.checkpoint_every_nbatches=null"]
from inverse_audio_synthesis_amd import optim
orig = optim.LARS.sync_hyper
def traced(self):
    orig(self)
    torch.cuda.synchronize()
    for gi, ent in self.__dict__.get("_hip_tables", {}).items():
        print("sync_hyper group", gi, "dev hyper", ent["hyper"].tolist(), "vals", ent["hyper_vals"], "ptr", hex(ent["hyper"].data_ptr()), flush=True)
optim.LARS.sync_hyper = traced
args = SMALL + ["trainer.max_steps=6", "param_embed.dropout=0.0"]
h_g = pretrain.app(args + ["trainer.cuda_graph=true", "trainer.out_dir=/tmp/dbg_g"])
optim.LARS.sync_hyper = orig
h_e = pretrain.app(args + ["trainer.cuda_graph=false", "trainer.out_dir=/tmp/dbg_e"])
for a, b in zip(h_e, h_g):
    print(a["step"], a["lr"], a["vicreg/train/loss"], b["vicreg/train/loss"])
