"""Per-tile phase timing of the audio-rate kernel from in-kernel s_memtime stamps (diagnostic build
scripts/diag/_bin/libias_stamps.so, -DVOICE_STAMPS).  Stamps go to a buffer of their own; no output depends on them."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["IAS_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libias_stamps.so")
import torch
from inverse_audio_synthesis_amd import _lib
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice

dev = torch.device("cuda:0")
B = 128
voice = Voice(SynthConfig(batch_size=B, reproducible=False)).to(dev)
voice.set_parameters01(torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev))
ws = voice.new_workspace(dev)
audio = torch.empty((B, voice.synthconfig.buffer_size), dtype=torch.float32, device=dev)
voice.render_control(ws)
V2 = True
TILE = 4096
ntiles = (voice.synthconfig.buffer_size + TILE - 1) // TILE
stamps = torch.zeros((B, ntiles, 4, 12), dtype=torch.int64, device=dev)
lib = ctypes.CDLL(os.environ["IAS_HIP_LIB"])
for _ in range(3):
    voice.render_audio(ws, out=audio, normalize=False)
torch.cuda.synchronize()
assert lib.ias_voice_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
voice.render_audio(ws, out=audio, normalize=False)
torch.cuda.synchronize()
st = stamps.cpu().double()[:, 1: ntiles - 1]          # full tiles, not the first of a row
names = ["issue + stage ctrl + barrier 1", "phase A (next tile)", "look-back (wave 0)", "barrier 2", "phase B + stores", "barrier 3"]
print("s_memtime ticks = shader cycles; one loop iteration = phase A of the next tile + look-back and phase B of this one; mean per wave")
if not V2:
    # wave tiles: a tile carries the stamps of the one wave that rendered it
    flat = st.reshape(-1, 12)
    flat = flat[flat[:, 0] > 0]
    names = ["put + ticket + DMA issue", "phase A (next tile) + publish", "look-back", "wait noise DMA", "phase B + stores", "-"]
    d = [(flat[:, i + 1] - flat[:, i]).mean().item() for i in range(6)]
    print("wave tiles: " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, d)) + f"; iteration {(flat[:, 6] - flat[:, 0]).mean().item():.0f} cycles over {flat.shape[0]} tiles")
    print(f"kernel span {(flat[:, 11].max() - flat[:, 10].min()).item() / 100:.1f} us")
    sys.exit(0)
for w in range(4):
    d = [(st[:, :, w, i + 1] - st[:, :, w, i]).mean().item() for i in range(6)]
    print(f"wave {w}: " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, d)) + f"; iteration {(st[:, :, w, 6] - st[:, :, w, 0]).mean().item():.0f} cycles")
cyc = (st[:, :, 0, 6] - st[:, :, 0, 0])
real = (st[:, :, 0, 11] - st[:, :, 0, 10])           # s_memrealtime: 100 MHz
print(f"in-kernel clock = {cyc.sum().item() / real.sum().item() * 0.1:.3f} GHz")
span_real = (st[:, :, :, 11].max() - st[:, :, :, 10].min()).item()
print(f"kernel span {span_real / 100:.1f} us")
