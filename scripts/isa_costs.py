#!/usr/bin/env python3
"""Static VALU cost per instruction of the step's three audio-rate kernels, from the ISA hipcc emits for them:
    python scripts/isa_costs.py            -> profiles/isa_costs.json
For each kernel the unrolled hot blocks (>= 200 instructions; all blocks inside loops if those are under half of the kernel) are classified and priced
with the cycle table measured by scripts/diag/valu_rates.hip / valu_occupancy.hip (SIMD clocks per wave64 instruction when
at least two waves of the SIMD have one ready): plain fp32 FMA/MUL/ADD/MOV 2, every other VALU instruction (fp64, packed
fp32, conversions, med3/min/max, compares, fract/rndne, integer multiplies, DPP moves, cndmask ...) 4, transcendentals 8.
bench.py multiplies `valu_clk_per_inst` with the SQ_INSTS_VALU count of profiles/counters.json to price a kernel's
vector-pipe time (`roofline.kernels.*.frac_valu`).  Needs hipcc only (no GPU)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "inverse-audio-synthesis_amd", "csrc")
BASE = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only"]
KERNELS = [  # (json key = the name bench.py / rocprofv3 use, source, extra flags as in csrc/Makefile, mangled-name substring)
    ("voice_audio_kernel", "voice_kernels.hip", ["-ffp-contract=off", "-fno-slp-vectorize"], "voice_audio_kernelILi0ELb1E"),
    ("stft2_kernel<8, true, 1, 1>", "spectral_kernels.hip", ["-fno-slp-vectorize"], "stft2_kernelILi8ELb1ELi1ELi1E"),
    ("pqmf_analysis_mods_kernel", "pqmf_kernels.hip", ["-fno-slp-vectorize"], "pqmf_analysis_mods_kernelILb0E"),
    # the configs[4] gradient step's transform kernels and the Voice backward (legs.gradstep.roofline.pipes)
    ("stft2_kernel<8, false, 2, 1>", "spectral_kernels.hip", ["-fno-slp-vectorize"], "stft2_kernelILi8ELb0ELi2ELi1E"),
    ("stft2_kernel<8, false, 2, 2>", "spectral_kernels.hip", ["-fno-slp-vectorize"], "stft2_kernelILi8ELb0ELi2ELi2E"),
    ("stft2h_kernel<8, 2>", "spectral_kernels.hip", ["-fno-slp-vectorize"], "stft2h_kernelILi8ELi2E"),
    ("stft_grad_wave_kernel<10, false, true>", "spectral_kernels.hip", ["-fno-slp-vectorize"], "stft_grad_wave_kernelILi10ELb0ELb1E"),
    ("stft_grad2k_kernel<8, true>", "spectral_kernels.hip", ["-fno-slp-vectorize"], "stft_grad2k_kernelILi8ELb1E"),
    ("stft_grad512_kernel<8>", "spectral_kernels.hip", ["-fno-slp-vectorize"], "stft_grad512_kernelILi8E"),
    ("voice_grad_sample16_kernel", "voice_grad_kernels.hip", ["-ffp-contract=off", "-fno-slp-vectorize"], "voice_grad_sample16_kernel"),
    ("voice_grad_pitch16_kernel", "voice_grad_kernels.hip", ["-ffp-contract=off", "-fno-slp-vectorize"], "voice_grad_pitch16_kernel"),
    ("pqmf_analysis_mfma_kernel<64, 63, 1, 2, 6>", "pqmf_kernels.hip", ["-fno-slp-vectorize"], "pqmf_analysis_mfma_kernelILi64ELi63ELi1ELi2ELi6E"),
    ("pqmf_synthesis_wide_kernel<64>", "pqmf_kernels.hip", ["-fno-slp-vectorize"], "pqmf_synthesis_wide_kernelILi64E"),
]
COST = {"fast": 2, "slow": 4, "f64": 4, "cvt": 4, "pk": 4, "trans": 8}


def cls_of(k):
    if not k.startswith("v_"):
        return None
    if k.startswith("v_cvt"): return "cvt"
    if "f64" in k: return "f64"
    if k.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if k.startswith("v_pk_"): return "pk"
    if k.startswith(("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_fmac_f32", "v_mac_f32",
                     "v_fmaak_f32", "v_fmamk_f32")) and "dpp" not in k: return "fast"
    return "slow"


def blocks_of(asm, key):
    m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", asm, re.M)
    if not m:
        raise SystemExit(f"kernel {key} not found")
    body = asm[m.start():asm.index(".Lfunc_end", m.start())]
    blocks, cur = [], []
    inloop = [False]
    for l in body.split("\n"):
        if re.match(r"^\.LBB\S+:", l):
            blocks.append(cur); cur = []
            inloop.append("in Loop" in l or "Loop Header" in l)
            continue
        if l.startswith("; %bb.") and ("in Loop" in l or "Loop Header" in l) and not cur:
            inloop[-1] = True
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        cur.append(t.split()[0])
    blocks.append(cur)
    return blocks, inloop


def main():
    out = {"_method": __doc__.split("\n\n")[0].replace("\n", " ") + "  Cycle table: plain fp32 2, other VALU 4, transcendental 8.",
           "_cost_table_clocks": COST}
    with tempfile.TemporaryDirectory() as td:
        for name, src, flags, key in KERNELS:
            asm_path = os.path.join(td, src + ".s")
            if not os.path.exists(asm_path):
                subprocess.run(BASE + flags + [os.path.join(CSRC, src), "-o", asm_path], check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, cwd=CSRC)
            try:
                blocks, inloop = blocks_of(open(asm_path).read(), key)
            except SystemExit as e:
                print("skipped:", e)
                continue
            hot = [b for b in blocks if len(b) >= 200]
            if sum(len(b) for b in hot) < 0.5 * sum(len(b) for b in blocks):
                # no dominant unrolled blocks: the static mix of the kernel's loops (everything if it has none)
                hot = [b for b, il in zip(blocks, inloop) if il] or blocks
            c = collections.Counter()
            for b in hot:
                for ins in b:
                    k = cls_of(ins)
                    if k:
                        c[k] += 1
            valu = sum(c.values())
            clk = sum(COST[k] * v for k, v in c.items())
            out[name] = {"valu_clk_per_inst": round(clk / valu, 3), "hot_blocks": len(hot), "hot_block_instructions": sum(len(b) for b in hot),
                         "hot_valu_by_class": dict(sorted(c.items())),
                         "hot_lds_instructions": sum(1 for b in hot for i in b if i.startswith("ds_"))}
            print(name, out[name])
    json.dump(out, open(os.path.join(ROOT, "profiles", "isa_costs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
