#!/usr/bin/env python3
"""SQ counter summaries of the step's three audio-rate kernels -> one JSON that bench.py imports:
    python scripts/make_counters.py <dir with pmc_voice.txt pmc_stft.txt pmc_pqmf.txt> <round tag> [out.json]
The summaries are what scripts/diag/pmc_{voice,stft,pqmf}.sh print (rocprofv3 --pmc passes, averages per launch).  The
per-launch values are chip-wide sums (SQ_*: over all CUs; GRBM_GUI_ACTIVE: over the 8 XCDs)."""
import json
import os
import re
import sys

WANT = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
        "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
PICK = {"pmc_voice.txt": ("voice_audio_kernel", "voice_audio_kernel"),
        "pmc_stft.txt": ("stft2_kernel<8, true, 1, 1>", "stft2_kernel<8, true, 1, 1>"),
        "pmc_pqmf.txt": ("pqmf_analysis_mods_kernel", "pqmf_analysis_mods_kernel")}


HASHED = ("voice_kernels.hip", "voice_math.h", "voice_trig.h", "wave_ops.h", "voice_exp2_table.h", "stft2_kernels.hip",
          "spectral_kernels.hip", "pqmf_kernels.hip", "ias_common.h", "Makefile")


def kernel_sources_sha16(root):
    """sha256 (first 16 hex digits) over the sources and build flags of the three profiled kernels: bench.py recomputes it
    and marks the imported counters stale when the kernels have changed since they were profiled."""
    import hashlib
    h = hashlib.sha256()
    for f in HASHED:
        with open(os.path.join(root, "inverse-audio-synthesis_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:16]


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if line[:1] not in (" ", "\n") and " calls " not in line:
            cur = line.strip()
            out.setdefault(cur, {})
        else:
            m = re.match(r"^\s+(\w+)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(2))
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(src, "counters.json")
    res = {"_method": "rocprofv3 --pmc passes of scripts/diag/pmc_voice.sh / pmc_stft.sh / pmc_pqmf.sh (each counter set in a run of its "
                      "own, --kernel-trace only), averages per launch at B = 128 x 176400; SQ_* are sums over all CUs, "
                      "GRBM_GUI_ACTIVE over the 8 XCDs", "_round": tag,
           "_source_sha16": kernel_sources_sha16(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}
    for fname, (needle, key) in PICK.items():
        p = os.path.join(src, fname)
        if not os.path.exists(p):
            continue
        for kname, vals in parse(p).items():
            if needle in kname and "SQ_INSTS_VALU" in vals:
                res[key] = {c: vals[c] for c in WANT if c in vals}
                res[key]["_rocprof_name"] = kname
                break
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
