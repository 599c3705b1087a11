import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import spectral_oracle as spo
from inverse_audio_synthesis_amd.spectral import STFTPlan, MelSpectrogram, VALUE_POWER
dev=torch.device('cuda:0')
g=torch.Generator().manual_seed(5)
x=torch.randn(3,20000,generator=g)*0.5
ref=spo.spectrogram(x,1024,None,512,2.0)
plan=STFTPlan(1024,None,512).to(dev)
out=plan.values(x.to(dev),VALUE_POWER).transpose(1,2).cpu()
err=(out-ref).abs()
print('raw max err', err.max().item(), 'scale', ref.abs().max().item(), 'worst bin', err.amax(dim=(0,2)).argmax().item(), 'worst frame', err.amax(dim=(0,1)).argmax().item())
bad=(err.amax(dim=(0,2))>1e-4*ref.abs().max()).nonzero().flatten().tolist()
print('bad bins', bad[:40], len(bad))
for sr in (16000,44100):
    refm=spo.mel_spectrogram(x,sample_rate=sr)
    mel=MelSpectrogram(sample_rate=sr).to(dev)
    om=mel(x.to(dev)).cpu()
    e=(om-refm).abs()
    print(sr,'mel max err', e.max().item(), 'scale', refm.abs().max().item())
    badm=(e.amax(dim=(0,2))>1e-4*refm.abs().max()).nonzero().flatten().tolist()
    print('  bad mels', badm[:40], len(badm), 'segtab', mel.plan.segtab is not None)
