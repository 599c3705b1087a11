#!/usr/bin/env python3
"""Headline benchmark: audio-seconds rendered+lossed per second (BASELINE.json metric).

One step = one pass of the hot path over one batch resident in HBM:
    Voice render (78 params -> [B,T] audio)  ->  PQMF(3) analysis  ->  mel-spectrogram L1 vs a cached
    target mel, at BASELINE config #2: batch 128 x 4 s @ 44.1 kHz per GPU (weak scaling: every rank
    renders its own batch, seeds 1000+rank / 2000+rank; no data-path collective).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line with the driver's contract fields plus "roofline" (dominant kernel,
HIP-event timed inside the timed region) and "cpu_baseline" (the oracle timed on the host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured float4-copy rate
SAMPLE_RATE, SECONDS, BATCH = 44100, 4.0, 128
# Algorithmic HBM bytes per audio sample: render = 4 B noise read + 4 B audio write.
RENDER_BYTES_PER_SAMPLE = 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 128; gradstep: 64)")
    ap.add_argument("--workload", default="synth", choices=["synth", "vicreg", "gradstep"],
                    help="synth: the headline (BASELINE configs[1]: render + PQMF(3) + mel-L1, forward); "
                         "vicreg: VICReg.loss forward + backward on [B, 8192] embeddings (configs[2]; with N > 1 ranks "
                         "the FullGatherLayer all-gather / reduce-scatter and the global-batch loss of configs[3]); "
                         "gradstep: configs[4] per GPU: render + 64-band PQMF + 3-resolution MR-STFT loss, forward + backward")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="finish a step's PQMF / spectral loss before the next step's render starts")
    ap.add_argument("--buffers", type=int, default=3, help="audio buffers / workspaces in flight (pipeline depth)")
    ap.add_argument("--no-defer-consumers", action="store_true",
                    help="issue a step's PQMF / STFT right after its render instead of after the next render "
                         "(the round-1 issue order; see run_steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the short vicreg / gradstep legs behind the headline")
    ap.add_argument("--cpu-batch", type=int, default=32, help="voices in the CPU baseline sample")
    ap.add_argument("--leg-child", action="store_true",
                    help="(internal) one rank of the N > 1 legs: started by the ranks of a `--gpus N` run after their headline "
                         "regions, on a process group of its own; rank 0 prints the legs' JSON")
    ap.add_argument("--replays", type=int, default=0,
                    help="timed regions of K steps each (0: as many as give >= 0.25 s of timed work, at least 20)")
    return ap.parse_args()


COLL_DEV = None   # device of the timing collectives' tensors (set in main)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU, RCCL
    rendezvous on 127.0.0.1) BEFORE this process touches the GPU, wait for them, return the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(abs(p.wait()) for p in procs)


def cpu_baseline(cpu_batch):
    """The oracle (torch-CPU restatement, math="torch": the ops the reference would issue) timed on the
    host cores over a bounded sample of the same workload: render + PQMF(3) + mel-L1 of cpu_batch voices.
    Up to three thread counts are timed (16 / 32 / 64, capped by the host: torch's CPU ops on [32, 176400] tensors stop
    scaling well before that, and on all 256 host threads the same sample ran 280x slower: 2.9 audio-s/s, round-2
    measurement); the fastest is the reported value, all are named in `sample`."""
    from oracle import pqmf_oracle as po
    from oracle import spectral_oracle as spo
    from oracle import synth_oracle as so
    cores = os.cpu_count() or 1
    cfg = so.VoiceConfig(batch_size=cpu_batch, sample_rate=SAMPLE_RATE, buffer_size_seconds=SECONDS)
    noise = so.make_noise(cfg)
    g = torch.Generator().manual_seed(1000)
    params = torch.rand(cpu_batch, 78, generator=g)
    tgt = so.render_from_params01(cfg, torch.rand(cpu_batch, 78, generator=torch.Generator().manual_seed(2000)),
                                  noise, "torch")
    tgt_mel = spo.mel_spectrogram(tgt, sample_rate=SAMPLE_RATE)
    H, _, _ = po.design(3)

    def step():
        audio = so.render_from_params01(cfg, params, noise, "torch")
        z = po.analysis(audio.unsqueeze(1), H, 3, 62)
        loss = torch.mean(torch.abs(spo.mel_spectrogram(audio, sample_rate=SAMPLE_RATE) - tgt_mel))
        return z, loss

    runs = []
    for threads in sorted({min(cores, 16), min(cores, 32), min(cores, 64)}):
        torch.set_num_threads(threads)
        step()  # warm-up (thread pool, allocator)
        reps, t0 = 0, time.perf_counter()
        while True:
            step()
            reps += 1
            el = time.perf_counter() - t0
            if (el > 6.0 and reps >= 5) or reps >= 200 or el > 30.0:
                break
        runs.append((cpu_batch * SECONDS * reps / el, threads, reps, el))
    best = max(runs)
    return {
        "value": round(best[0], 2),
        "unit": "audio-s/s",
        "cores": best[1],
        "host_threads": cores,
        "kind": "port",
        "sample": f"render+PQMF(3)+mel-L1 over {cpu_batch} voices x {SECONDS:g} s @ {SAMPLE_RATE} Hz (oracle, torch CPU ops): "
                  + "; ".join(f"{r[2]} passes in {r[3]:.1f} s on {r[1]} threads = {r[0]:.1f} audio-s/s" for r in runs),
    }


MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)


def timed_regions(one_region_fn, args, world, dev):
    """R timed regions of exactly K steps each (barrier + synchronize on both sides, max over ranks); -> sorted list."""
    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def one():
        sync_all()
        t0 = time.perf_counter()
        one_region_fn()
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=COLL_DEV)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el

    first = one()
    replays = args.replays if args.replays > 0 else max(20, min(2000, int(0.25 / max(first, 1e-6)) + 1))
    if world > 1:
        r = torch.tensor([replays], dtype=torch.int64, device=COLL_DEV)
        dist.broadcast(r, 0)
        replays = int(r.item())
    return sorted(one() for _ in range(replays))


def vicreg_cpu_baseline(B, D):
    """oracle/vicreg_oracle.py (the reference's op sequence, vicreg.py:35-58) forward + autograd backward on the host."""
    from oracle import vicreg_oracle as vo
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(0)).requires_grad_()
    y = torch.randn(B, D, generator=torch.Generator().manual_seed(1)).requires_grad_()

    def step():
        x.grad = y.grad = None
        vo.loss(x, y, B, D)[0].backward()

    step()
    reps, t0 = 0, time.perf_counter()
    while True:
        step()
        reps += 1
        el = time.perf_counter() - t0
        if (el > 8.0 and reps >= 3) or el > 40.0:
            break
    return {"value": round(B * reps / el, 2), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{reps} passes of VICReg.loss forward + backward on x, y [{B}, {D}] (oracle, torch CPU ops, {el:.1f} s)"}


def vicreg_roofline(batch_side, kernel, achieved, gram_ms, executed, nominal, kpad, step_bytes, step_s, dxd):
    """The vicreg workload's roofline object.  Batch-side form (the product path where the padded batch <= D): the step is
    six small launches moving x, y in and gx, gy out -- bound "hbm" on those bytes (a small problem: the fraction says how
    far from streaming it is), with the B x B contraction's matrix-core numbers under "contraction" and the D x D kernel of
    rounds 1-3, timed in the same run, under "dxd".  Feature-side form: the D x D Gram kernel against the bf16 MFMA peak,
    as in rounds 1-3."""
    contraction = {"kernel": kernel, "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "avg_launch_ms": round(gram_ms, 4), "flops_executed": executed,
                   "flops_nominal_2BD2_per_branch_x2": nominal,
                   "measured": "HIP events around K back-to-back launches of stage 1 (the covariance contraction) on the "
                               "global batch, fastest of 15 replays"}
    if not batch_side:
        contraction["form"] = "feature side (D x D)"
        contraction["frac_nominal"] = round(nominal / (gram_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
        contraction["traffic"] = pmc_traffic_of("vicreg_gram_pair_kernel" if kpad == 128 else "vicreg_gram256_kernel")
        return contraction
    gbps = step_bytes / step_s / 1e9
    return {"kernel": "VICReg.loss forward + backward, batch-side form: colstats, B x B product, fold, finish | gradient product",
            "form": "batch side (B x B)", "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbps / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_step": step_bytes,
            "what": "x, y read, gx, gy written (fp32), per rank, over the step's time (driver-timed)",
            "contraction": contraction, "dxd": dxd}


def run_vicreg(args, rank, world, dev):
    """BASELINE configs[2] / [3]: the projector loss on [B, 8192] embeddings, forward + backward, through
    ``vicreg.global_batch_loss`` -- the function ``VICReg.loss`` calls.  N = 1: the local loss.  N > 1: cat(x, y) all-gathered
    over RCCL, the loss on the global batch with denominator B_global - 1, the gradient reduce-scattered back -- one real
    exchange step per direction."""
    from inverse_audio_synthesis_amd import _lib
    from inverse_audio_synthesis_amd.vicreg import global_batch_loss
    lib = _lib.load()
    B, D = args.batch or BATCH, 8192
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(2 * rank)).to(dev).requires_grad_()
    y = torch.randn(B, D, generator=torch.Generator().manual_seed(2 * rank + 1)).to(dev).requires_grad_()
    gather = world > 1
    state = {}
    one = torch.ones((), dtype=torch.float32, device=dev)

    def step():
        # the code VICReg.loss runs (inverse-audio-synthesis_amd/vicreg.py): N > 1 = ONE all-gather of cat(x, y, dim=1), the
        # loss on the gathered buffer in place with denominator B * world - 1, one reduce-scatter of its cotangent
        out = global_batch_loss(x, y, B, 25.0, 25.0, 1.0, gather=True)
        # `one`: the loss' cotangent as a trainer passes it (autograd otherwise fills a fresh 1 per step: a 5 us launch)
        gx, gy = torch.autograd.grad(out[0], (x, y), one)     # backward of the loss (and of the gather)
        state["out"] = tuple(v.detach() for v in out)
        state["grads"] = (gx, gy)

    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    launch, graph = "eager", None
    # the K steps replay from one hipGraph -- with the gather too when the collectives are RCCL's (they capture like any
    # other stream work; a gloo rehearsal stays eager)
    if not args.no_graph and (not gather or dist.get_backend() == "nccl"):
        try:
            state.clear()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if gather else "global"):
                for _ in range(args.steps):
                    step()
            graph.replay()
            torch.cuda.synchronize()
            launch = "hipgraph"
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly\n")
            graph = None
            torch.cuda.synchronize()
            if not state:
                step()

    def region():
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                step()

    regions = timed_regions(region, args, world, dev)
    elapsed = regions[len(regions) // 2]
    out = [float(v) for v in state["out"]]

    # ---- the covariance contraction alone (stage 1 of ias_vicreg_stage on a filled workspace), HIP events around K launches.
    # The product path contracts over the batch where the padded batch <= D (B x B matrix, 2 B^2 D flops: SURVEY 8(d)'s
    # identity); the feature-side D x D kernels of rounds 1-3 (2 B D^2 flops) are timed next to it through the diagnostic
    # library under ias_vicreg_set_form(0), and the whole forward + backward once more in that form: `roofline.dxd`.
    Bg = B * world
    xg = torch.randn(Bg, D, generator=torch.Generator().manual_seed(100)).to(dev)
    yg = torch.randn(Bg, D, generator=torch.Generator().manual_seed(101)).to(dev)
    need = int(lib.ias_vicreg_workspace_bytes(Bg, D))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    o4 = torch.empty(4, dtype=torch.float32, device=dev)

    def stage(k):
        _lib.check(lib.ias_vicreg_stage(k, _lib.ptr(xg), _lib.ptr(yg), _lib.ptr(o4), _lib.ptr(ws), need, Bg, D, Bg, 25.0, 25.0,
                                        1.0, _lib.stream()), "ias_vicreg_stage")

    def time_graph(fn, reps=15):
        """-> ms per call of fn: K calls captured into one graph, the fastest of `reps` replays (short bursts)"""
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(args.steps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.steps)
        return best

    kpad = (Bg + 127) // 128 * 128
    ntile = (D + 127) // 128
    batch_side = kpad <= D and D % 8 == 0
    stage(-1)
    gram_ms = time_graph(lambda: stage(1))
    loss_default = o4.tolist()
    nominal = 2.0 * 2.0 * Bg * D * D                                          # 2 B D^2 per branch (vicreg.py:47-48)
    # feature side: both branches, the upper triangle at 128 x 128 granularity (the 256 x 256 kernel executes 1.5 % more)
    executed_dxd = 2.0 * (ntile * (ntile + 1) // 2) * 2.0 * 128 * 128 * kpad
    dxd_name = ("vicreg_gram_pair_kernel" if kpad == 128 else
                ("vicreg_gram_kernel (128 x 128 tiles; both branches, one launch)" if D < 256 else
                 "vicreg_gram256_kernel (256 x 256 tiles, LDS-DMA; both branches, one launch)"))
    dxd = None
    if batch_side:
        t256 = kpad > 128
        bt = (kpad + 255) // 256 if t256 else kpad // 128
        tile = 256 if t256 else 128
        executed = 2.0 * (bt * (bt + 1) // 2) * 2.0 * tile * tile * D         # upper-triangular tiles of Xc Xc^T, both branches
        kernel = ("vicreg_bgram256_kernel + vicreg_gconv256_kernel" if t256 else "vicreg_bgram_kernel + vicreg_gconv_kernel") + \
                 " (Xc Xc^T, B x B, in D-slices + the fixed-order fold that also sums its squares)"
        # the D x D kernels on the same inputs, and the whole step in that form: through the DIAGNOSTIC library (the product
        # library has no switch -- the side follows from the shape; include/ias_hip_diag.h), a separate library instance
        diag = _lib.load_diag()
        _lib.check(diag.ias_vicreg_set_form(0), "ias_vicreg_set_form")
        try:
            with _lib.use_library(diag):
                need0 = int(diag.ias_vicreg_workspace_bytes(Bg, D))
                ws0 = torch.empty(need0, dtype=torch.uint8, device=dev)

                def stage0(k):
                    _lib.check(diag.ias_vicreg_stage(k, _lib.ptr(xg), _lib.ptr(yg), _lib.ptr(o4), _lib.ptr(ws0), need0, Bg, D, Bg,
                                                     25.0, 25.0, 1.0, _lib.stream()), "ias_vicreg_stage (diagnostic library)")
                stage0(-1)
                dxd_ms = time_graph(lambda: stage0(1))
                loss_dxd = o4.tolist()
                step_dxd_ms = None
                if not gather:
                    step_dxd_ms = time_graph(step, reps=7)
        finally:
            _lib.check(diag.ias_vicreg_set_form(-1), "ias_vicreg_set_form")
        a_dxd = executed_dxd / (dxd_ms * 1e-3) / 1e12
        dxd = {"kernel": dxd_name, "note": "the feature-side contraction of rounds 1-3: NOT in the product path at this shape (the "
                                           "product library picks the side from the shape); timed through libias_hip_diag.so "
                                           "(ias_vicreg_set_form(0)) for continuity",
               "avg_launch_ms": round(dxd_ms, 4), "flops_executed": executed_dxd, "achieved": round(a_dxd, 1),
               "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(a_dxd / MFMA_BF16_PEAK_TFLOPS, 4),
               "traffic": pmc_traffic_of("vicreg_gram_pair_kernel" if kpad == 128 else "vicreg_gram256_kernel"),
               "step_ms_in_this_form": round(step_dxd_ms, 4) if step_dxd_ms else None,
               "cov_loss": loss_dxd[3], "cov_loss_batch_side": loss_default[3]}
    else:
        executed, kernel = executed_dxd, dxd_name
    achieved = executed / (gram_ms * 1e-3) / 1e12
    step_bytes = 16.0 * B * D                                                 # read x, y; write gx, gy (fp32), per rank
    result = {
        "metric": "VICReg.loss forward+backward, embeddings [B, 8192] per GPU (BASELINE configs[2]; N>1: configs[3] with "
                  "the FullGatherLayer all-gather / reduce-scatter over RCCL)",
        "value": round(world * B * args.steps / elapsed, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "ms_per_step_min": round(regions[0] / args.steps * 1e3, 4), "timed_regions": len(regions),
        "timed_region_s": round(elapsed, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 MFMA Gram (fp32 centring and accumulation), fp32 elsewhere", "data": "synthetic",
        "config": {"workload": f"VICReg.loss fwd+bwd, x, y = randn({B}, {D}) per rank, global batch {Bg}",
                   "batch_per_gpu": B, "embeddim": D, "launch": launch, "gather": gather,
                   "collective": "all_gather_into_tensor fwd + reduce_scatter_tensor bwd (RCCL)" if gather else None,
                   "rccl_world_size": dist.get_world_size() if gather else 1,
                   "loss": out[0], "repr_loss": out[1], "std_loss": out[2], "cov_loss": out[3]},
        "roofline": vicreg_roofline(batch_side, kernel, achieved, gram_ms, executed, nominal, kpad, step_bytes, elapsed / args.steps, dxd),
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = vicreg_cpu_baseline(B, D)
    return result


def secondary_legs(args, dev):
    """Short runs of the other workloads (bench.py --workload vicreg / gradstep) so that ONE default invocation carries
    a driver-timed MFMA number and a configs[4] number: -> {"vicreg128": {...}, "vicreg1024": {...}, "gradstep": {...}}"""
    import copy
    legs = {}
    for name, wl, batch, steps in (("vicreg128", "vicreg", 128, 20), ("vicreg1024", "vicreg", 1024, 10),
                                   ("gradstep", "gradstep", 64, 5)):
        a2 = copy.copy(args)
        a2.workload, a2.batch, a2.steps, a2.warmup, a2.replays, a2.no_cpu_baseline = wl, batch, steps, 2, 10, True
        t0 = time.perf_counter()
        try:
            r = (run_vicreg if wl == "vicreg" else run_gradstep)(a2, 0, 1, dev)
            leg = {"ms_per_step": r["ms_per_step"], "ms_per_step_min": r["ms_per_step_min"], "steps": steps,
                   "timed_regions": r["timed_regions"], "value": r["value"], "unit": r["unit"],
                   "workload": r["config"]["workload"], "launch": r["config"]["launch"]}
            rf = r["roofline"]
            leg["roofline"] = {k: rf[k] for k in ("kernel", "form", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                  "frac_nominal", "flops_executed", "algorithmic_bytes_per_step", "contraction",
                                                  "dxd", "pipes") if k in rf}
        except Exception as e:  # noqa: BLE001 -- a failing leg must not take the headline line down
            leg = {"error": f"{type(e).__name__}: {e}"}
        leg["wall_s"] = round(time.perf_counter() - t0, 2)
        legs[name] = leg
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    for name, fn in (("pretrain", lambda: pretrain_leg(dev)), ("config0", lambda: config0_leg(dev, cpu=not args.no_cpu_baseline))):
        t0 = time.perf_counter()
        try:
            leg = fn()
        except Exception as e:  # noqa: BLE001
            leg = {"error": f"{type(e).__name__}: {e}"}
        leg["wall_s"] = round(time.perf_counter() - t0, 2)
        legs[name] = leg
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return legs


def pretrain_leg(dev, B=128, steps=5, reps=5):
    """One VICReg pretraining step of BASELINE configs[2] per GPU (B=128 x 4 s @ 44.1 kHz, dim 1024, embeddim 8192): render
    + PQMF + MobileNetV3 trunk + projector + VICReg loss + backward + LARS, as the Trainer's captured hipGraph
    (trainer.cuda_graph=true): the SURVEY 8(f) rows in one number.  Parameters are sampled on the host per step and copied
    in, everything else replays.  -> {"ms_per_step": median over `reps` regions of `steps` replays, ...}"""
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import Trainer
    import warnings
    torch.manual_seed(42)
    cfg = load_config(os.path.join(ROOT, "conf"), "config", [f"vicreg.batch_size={B}", "trainer.cuda_graph=true"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = VicregAudioParams(cfg)
    tr = Trainer(cfg, model, stage="vicreg", device=dev)
    model.train()
    for i in range(5):                       # 3 eager warm-up steps, capture, first replays
        tr._graph_step(i, i)
    torch.cuda.synchronize()
    times = []
    k = 10
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            tr._graph_step(k, k)
            k += 1
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / steps)
    times.sort()
    loss = float(model.logged["vicreg/train/loss"])
    ms = times[len(times) // 2]
    gemm_tuning = tr.gemm_tuning
    del tr, model
    return {"ms_per_step": round(ms, 3), "ms_per_step_min": round(times[0], 3), "steps": steps, "timed_regions": reps,
            "value": round(B * SECONDS / (ms * 1e-3), 1), "unit": "audio-s/s trained",
            "workload": f"VICReg pretraining step, batch {B} x {SECONDS:g} s @ {SAMPLE_RATE} Hz per GPU (configs[2] shape): render + "
                        "PQMF(3) + MobileNetV3-small trunk + projector 8192 + VICReg loss, backward, LARS; fp32 (bf16 only "
                        "inside the VICReg Gram), random init, synthetic parameters",
            "launch": "hipgraph (Trainer._graph_step)", "loss": loss, "gemm_tuning": gemm_tuning}


def config0_leg(dev, cpu=True):
    """BASELINE configs[0] -- the reference's own CPU-runnable case: one Voice, batch 4, 1 s @ 16 kHz, STFT-L1 loss (no
    gradient: audio_to_params.py's test path) -- timed on the HIP path (K steps in a hipGraph) and, same inputs, on the
    oracle's torch-op restatement on the host: legs.config0."""
    from inverse_audio_synthesis_amd.spectral import STFTL1
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    B, sr, sec, K = 4, 16000, 1.0, 50
    cfg = SynthConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec, reproducible=False)
    voice = Voice(cfg).to(dev)
    loss_mod = STFTL1().to(dev)
    p = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000))
    pt = torch.rand(B, 78, generator=torch.Generator().manual_seed(2000))
    tgt = voice.render(pt.to(dev)).clone()
    pd = p.to(dev)
    out = {}

    def step():
        out["loss"] = loss_mod(voice.render(pd), tgt)

    step(); step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            step()
    g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / K)
    times.sort()
    leg = {"workload": f"BASELINE configs[0]: Voice render + STFT-L1 (n_fft 1024, hop 512), batch {B} x {sec:g} s @ {sr} Hz",
           "ms_per_step": round(times[len(times) // 2], 5), "ms_per_step_min": round(times[0], 5), "steps": K,
           "launch": "hipgraph", "value": round(B * sec / (times[len(times) // 2] * 1e-3), 1), "unit": "audio-s/s",
           "loss": float(out["loss"])}
    if cpu:
        from oracle import spectral_oracle as spo
        from oracle import synth_oracle as so
        ocfg = so.VoiceConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec)
        noise = so.make_noise(ocfg)
        tgt_c = so.render_from_params01(ocfg, pt, noise, "torch")
        threads = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(threads)

        def cpu_step():
            return spo.stft_l1(so.render_from_params01(ocfg, p, noise, "torch"), tgt_c)
        cpu_step()
        reps, t0 = 0, time.perf_counter()
        while True:
            ref = cpu_step()
            reps += 1
            el = time.perf_counter() - t0
            if (el > 2.0 and reps >= 5) or el > 10.0:
                break
        cpu_ms = el / reps * 1e3
        leg["cpu_baseline"] = {"value": round(B * sec / (cpu_ms * 1e-3), 1), "unit": "audio-s/s", "ms_per_step": round(cpu_ms, 3),
                               "cores": threads, "kind": "port",
                               "sample": f"{reps} passes of the same step (oracle, torch CPU ops) in {el:.1f} s", "loss": float(ref)}
        leg["loss_rel_diff_vs_oracle"] = abs(leg["loss"] - float(ref)) / max(abs(float(ref)), 1e-30)
    return leg


# ---------------------------------------------------------------------------------------------- N > 1 legs (round 5)
LEG_PG_TIMEOUT_S, LEG_WALL_TIMEOUT_S = 150, 420


def vicreg_gather_leg(args, rank, world, dev):
    """legs.vicreg_gather: configs[3]'s loss per rank -- global_batch_loss on B_l = 128 rows per rank, ONE all-gather of
    cat(x, y) forward and ONE reduce-scatter backward over RCCL (reference: vicreg.py:38-39,79-95)."""
    import copy
    a2 = copy.copy(args)
    a2.workload, a2.batch, a2.steps, a2.warmup, a2.replays, a2.no_cpu_baseline = "vicreg", 128, 10, 2, 10, True
    r = run_vicreg(a2, rank, world, dev)
    leg = {k: r[k] for k in ("ms_per_step", "ms_per_step_min", "steps", "timed_regions", "value", "unit")}
    leg.update(workload=r["config"]["workload"], launch=r["config"]["launch"], collective=r["config"]["collective"],
               rccl_world_size=r["config"]["rccl_world_size"], backend=dist.get_backend(), loss=r["config"]["loss"],
               cov_loss=r["config"]["cov_loss"],
               gathered_bytes_per_rank=int(2 * 8192 * 4 * 128 * world), roofline={k: r["roofline"][k] for k in
                                                                                   ("bound", "achieved", "peak", "unit", "frac") if k in r["roofline"]})
    return leg


def pretrain_ddp_leg(dev, rank, world, B=128, steps=5, reps=3):
    """legs.pretrain_ddp: the configs[2] pretraining step on every rank with the bucketed gradient all-reduce
    (dist.GradBucketer; reference: Lightning strategy "ddp", conf/config.yaml:6-8, pretrain.py:97-99) -- eager, as one
    captured hipGraph with the collectives inside (RCCL only), and the same step WITHOUT the collectives: the difference is
    the all-reduce time the step does not hide."""
    import warnings
    from inverse_audio_synthesis_amd import dist as ias_dist
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import Trainer
    gloo = dist.get_backend() != "nccl"
    if gloo:
        steps, reps = 2, 1          # (rehearsal: the 570 MB of gradients cross the host)

    def build(local_only, graph):
        torch.manual_seed(42)
        cfg = load_config(os.path.join(ROOT, "conf"), "config", [f"vicreg.batch_size={B}", f"trainer.cuda_graph={'true' if graph else 'false'}"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = VicregAudioParams(cfg)
        tr = Trainer(cfg, model, stage="vicreg", device=dev)
        if local_only:
            tr.bucketer.close()        # (the Trainer's own: its hooks would go on filling and reducing its buckets)
            tr.bucketer = ias_dist.GradBucketer(model, bucket_bytes=int(cfg.trainer.bucket_mb) << 20, local_only=True)
        model.train()
        return tr, model

    def timed(fn):
        """median ms per step over `reps` regions of `steps` steps, max over ranks"""
        ts = []
        k = 100
        for _ in range(reps):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn(k * world + rank, k)
                k += 1
            torch.cuda.synchronize()
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=COLL_DEV)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            ts.append(el.item() / steps * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    leg = {"workload": f"VICReg pretraining step (configs[2] shape, batch {B} per rank) x {world} ranks, gradients averaged by "
                       "dist.GradBucketer (bucketed all-reduce overlapped with backward)", "steps": steps, "timed_regions": reps,
           "rccl_world_size": dist.get_world_size(), "backend": dist.get_backend()}
    tr, model = build(False, False)
    nbytes = sum(f.numel() * f.element_size() for f, _ in tr.bucketer.buckets)
    leg.update(gradient_bytes=int(nbytes), buckets=len(tr.bucketer.buckets), collective=tr.bucketer.collective)
    for i in range(3):
        tr._eager_step(i * world + rank, i)
    leg["ms_per_step_eager"] = round(timed(tr._eager_step), 3)
    leg["loss"] = float(model.logged["vicreg/train/loss"])
    del tr, model
    torch.cuda.empty_cache()
    if not gloo:
        tr, model = build(False, True)
        if tr._use_graph():
            for i in range(5):
                tr._graph_step(i * world + rank, i)
            if getattr(tr, "_graph", None) is not None:
                leg["ms_per_step_graph"] = round(timed(tr._graph_step), 3)
            else:
                leg["graph_error"] = "capture failed: see stderr"
        del tr, model
        torch.cuda.empty_cache()
    # the same step without the collectives (every rank for itself)
    tr, model = build(True, not gloo)
    if not gloo and tr._use_graph():
        for i in range(5):
            tr._graph_step(i * world + rank, i)
        leg["ms_per_step_no_allreduce"] = round(timed(tr._graph_step), 3)
        ref = leg.get("ms_per_step_graph")
    else:
        for i in range(3):
            tr._eager_step(i * world + rank, i)
        leg["ms_per_step_no_allreduce"] = round(timed(tr._eager_step), 3)
        ref = leg["ms_per_step_eager"]
    if ref is not None:
        leg["allreduce_exposed_ms"] = round(ref - leg["ms_per_step_no_allreduce"], 3)
        leg["allreduce_exposed_note"] = "step with the bucketed all-reduce minus the same step without it (same launch mode)"
    best = leg.get("ms_per_step_graph") or leg["ms_per_step_eager"]
    leg.update(ms_per_step=best, value=round(world * B * SECONDS / (best * 1e-3), 1), unit="audio-s/s trained (whole job)")
    del tr, model
    torch.cuda.empty_cache()
    return leg


def gradstep_all_ranks_leg(args, rank, world, dev):
    """legs.gradstep_all_ranks: BASELINE configs[4] as it is stated -- global batch 64 x N (512 at N = 8), 64-band PQMF +
    3-resolution STFT loss, gradient step -- every rank its share at the same time, batch-split, no data-path collective;
    value = the whole job's audio-seconds over the slowest rank's time."""
    import copy
    a2 = copy.copy(args)
    a2.workload, a2.batch, a2.steps, a2.warmup, a2.replays, a2.no_cpu_baseline = "gradstep", 64, 5, 2, 10, True
    r = run_gradstep(a2, rank, world, dev)
    return {"ms_per_step": r["ms_per_step"], "ms_per_step_min": r["ms_per_step_min"], "steps": r["steps"],
            "timed_regions": r["timed_regions"], "value": r["value"], "unit": r["unit"], "n_gpus": world,
            "global_batch": 64 * world, "workload": r["config"]["workload"], "launch": r["config"]["launch"],
            "loss": r["config"]["loss"], "scaling": "weak"}


def run_leg_child(args):
    """One rank of the N > 1 legs (started by `multi_gpu_legs`): a process group of its own with a bounded timeout, so that
    a wedged collective ends THIS process and becomes the leg's `error` -- the headline line of the parent is already
    computed.  Rank 0 prints one JSON line per leg, {"leg": name, "result": {...}}, as soon as the leg is done."""
    import datetime
    world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("IAS_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=LEG_PG_TIMEOUT_S), **kw)
    global COLL_DEV
    COLL_DEV = dev if backend == "nccl" else torch.device("cpu")
    from inverse_audio_synthesis_amd import _lib
    _lib.load()
    for name, fn in (("vicreg_gather", lambda: vicreg_gather_leg(args, rank, world, dev)),
                     ("pretrain_ddp", lambda: pretrain_ddp_leg(dev, rank, world)),
                     ("gradstep_all_ranks", lambda: gradstep_all_ranks_leg(args, rank, world, dev))):
        t0 = time.perf_counter()
        try:
            leg = fn()
        except Exception as e:  # noqa: BLE001
            leg = {"error": f"{type(e).__name__}: {e}"[:500]}
        leg["wall_s"] = round(time.perf_counter() - t0, 2)
        if rank == 0:
            # one line per leg, as soon as it is done: a later leg that takes the process down (a collective that expires
            # ends the process, it does not raise) does not take the finished ones with it
            print(json.dumps({"leg": name, "result": leg}), flush=True)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    dist.barrier()
    dist.destroy_process_group()


def multi_gpu_legs(args, rank, world, local_rank):
    """After the headline's timed regions, at N > 1: every rank starts ONE child process (a fresh interpreter; this process
    has initialised the GPU, so it never execs) that joins a second process group on a port rank 0 picks, runs the legs
    that have a real exchange step -- legs.vicreg_gather (RCCL all-gather + reduce-scatter, configs[3]) and
    legs.pretrain_ddp (bucketed gradient all-reduce, eager and captured) -- and exits.  A child that dies or outlives its
    time limit becomes {"error": ...}; the headline line is not at risk."""
    import socket
    import subprocess
    port = torch.zeros(1, dtype=torch.int64, device=COLL_DEV)
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast(port, 0)
    # (under torch.distributed.run the environment says TORCHELASTIC_USE_AGENT_STORE=True: every rank, rank 0 included, would
    # look for the launcher's store on the new port instead of rank 0 hosting one -- the children get a plain env:// set-up)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(int(port.item())))
    torch.cuda.empty_cache()
    cmd = [sys.executable, os.path.abspath(__file__), "--leg-child", "--gpus", str(world)]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=LEG_WALL_TIMEOUT_S)
        rc, out, err = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        so = e.stdout or ""
        rc, out, err = -9, so.decode(errors="replace") if isinstance(so, bytes) else so, f"no result within {LEG_WALL_TIMEOUT_S} s"
    legs = None
    if rank == 0:
        legs = {}
        for ln in out.splitlines():
            if ln.startswith('{"leg"'):
                try:
                    rec = json.loads(ln)
                    legs[rec["leg"]] = rec["result"]
                except ValueError:
                    pass
        for name in ("vicreg_gather", "pretrain_ddp", "gradstep_all_ranks"):
            if name not in legs:
                legs[name] = {"error": f"legs child of rank 0 ended with code {rc} before this leg reported: {(err or '')[-400:]}"}
        legs["wall_s"] = round(time.perf_counter() - t0, 2)
    ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int64, device=COLL_DEV)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0 and int(ok.item()) == 0 and rc == 0:
        legs["note"] = "a child of another rank failed"
    return legs


def pmc_traffic_of(kname):
    """HBM-side bytes per launch of a kernel from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    scripts/refresh_profiles.sh), or None."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:  # noqa: BLE001
        return None
    hits = [v for k, v in table.items() if kname in k and isinstance(v, dict)]
    return hits[0].get("hbm_bytes_per_launch") if hits else None


def gradstep_pipes(step_ms, B):
    """What the gradient step's pipes are busy for, from profiles/counters_gradstep.json (scripts/diag/pmc_gradstep.sh: SQ
    counters per kernel and launch at B = 64) and profiles/isa_costs.json (static clocks per VALU instruction): per kernel
    the vector-pipe and LDS-pipe time it needs on the whole chip, and their sums as fractions of THIS run's step -- the
    evidence behind "the step is bound by the transforms' vector work, not by HBM" (verdict r04 item 6).  None when the
    counters are absent, were taken at another batch, or on other sources."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "counters_gradstep.json")))
        costs = json.load(open(os.path.join(ROOT, "profiles", "isa_costs.json")))
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from make_counters import kernel_sources_sha16
        stale = c.get("_source_sha16") != kernel_sources_sha16(ROOT)
    except Exception:  # noqa: BLE001
        return None
    if B != 64:
        return None
    n_cu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    clk_hz = 2.1e9
    rows, valu_us, lds_us = {}, 0.0, 0.0
    # launches per step: the profiled run also computed the loss targets once (their transforms are not part of a step)
    render_calls = next((v.get("calls", 0) for k, v in c.items() if isinstance(v, dict) and k.startswith("voice_audio_kernel")), 0)
    n_steps = max(render_calls - 1, 1)
    for k, v in c.items():
        if not isinstance(v, dict) or "SQ_INSTS_VALU" not in v:
            continue
        per_step = round(v.get("calls", 0) / n_steps)
        if per_step < 1:
            continue
        cost = next((x["valu_clk_per_inst"] for kk, x in costs.items() if isinstance(x, dict) and kk.split("<")[0] == k.split("<")[0]
                     and (kk == k or "<" not in kk)), None)
        cost = cost if cost is not None else next((x["valu_clk_per_inst"] for kk, x in costs.items()
                                                   if isinstance(x, dict) and kk.split("<")[0] == k.split("<")[0]), 3.5)
        v_us = per_step * v["SQ_INSTS_VALU"] * cost / (4 * n_cu) / clk_hz * 1e6
        l_us = per_step * v.get("SQ_LDS_IDX_ACTIVE", 0.0) / n_cu / clk_hz * 1e6
        valu_us += v_us
        lds_us += l_us
        if v_us + l_us >= 5.0:
            rows[k] = {"launches_per_step": per_step, "valu_us": round(v_us, 1), "lds_us": round(l_us, 1),
                       "in_step_avg_us": round(v.get("avg_us", 0.0), 1),
                       "valu_insts": int(v["SQ_INSTS_VALU"]), "valu_clk_per_inst": cost,
                       "lds_conflict_frac": round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 3)
                       if v.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in v else None}
    return {"valu_busy_us_per_step": round(valu_us, 1), "lds_busy_us_per_step": round(lds_us, 1),
            "valu_busy_frac_of_step": round(valu_us / (step_ms * 1e3), 4), "lds_busy_frac_of_step": round(lds_us / (step_ms * 1e3), 4),
            "counters_round": c.get("_round"), "counters_stale": stale,
            "note": "per kernel and launch: SQ_INSTS_VALU x static clocks per instruction / (4 SIMDs x CUs) and SQ_LDS_IDX_ACTIVE / CUs, "
                    "at 2.1 GHz = the time the kernel needs of the whole chip's vector / LDS pipes; sums over one step against this "
                    "run's step time.  Kernels below 5 us of pipe time are in the sums but not listed",
            "kernels": rows}


def run_gradstep(args, rank, world, dev):
    """BASELINE configs[4], one GPU's share (global batch 512 / 8): params -> Voice render -> {3-resolution MR-STFT loss,
    64-band PQMF sub-band L1} -> gradient w.r.t. the 78 normalised parameters, all HIP (render and its backward,
    spectral_kernels / spectral_grad_kernels, wide PQMF analysis and its adjoint).  Batch-split, no data-path collective."""
    from inverse_audio_synthesis_amd.pqmf import PQMF
    from inverse_audio_synthesis_amd.spectral import MultiResolutionSTFTLoss, ParallelLossSum, SubbandL1
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    B = args.batch or 64
    cfg = SynthConfig(batch_size=B, sample_rate=SAMPLE_RATE, buffer_size_seconds=SECONDS, reproducible=False)
    T = cfg.buffer_size
    voice = Voice(cfg).to(dev)
    gram = PQMF(N=64).to(dev)
    mr = MultiResolutionSTFTLoss().to(dev)
    sub = SubbandL1(gram)
    both = ParallelLossSum(mr, sub)          # the two loss branches on streams of their own, forward and backward
    both.parallel = os.environ.get("IAS_BENCH_SERIAL_LOSSES", "0") in ("", "0")
    if os.environ.get("IAS_BENCH_MR_SERIAL"):      # (diagnostics: the three STFT resolutions one after the other on one stream)
        mr.parallel = False
    if os.environ.get("IAS_BENCH_PLS_SHARE"):      # (diagnostics: which STFT resolution's stream the sub-band branch borrows)
        both.share_from = int(os.environ["IAS_BENCH_PLS_SHARE"])
    params = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000 + rank)).to(dev).requires_grad_(True)
    tgt = voice.render(torch.rand(B, 78, generator=torch.Generator().manual_seed(2000 + rank)).to(dev)).clone()
    tb, tm = sub.target(tgt), mr.target(tgt)
    state = {}
    one = torch.ones((), dtype=torch.float32, device=dev)     # the loss' cotangent, as a trainer passes it (no fill launch per step)

    def step(ev=None):
        if ev: ev[0].record()
        a = voice.render(params)
        if ev: ev[1].record()
        loss = both(a, [dict(targets=tm), dict(target_bands=tb)])
        if ev: ev[2].record()
        (g,) = torch.autograd.grad(loss, params, one)
        if ev: ev[3].record()
        state["loss"], state["g"] = loss.detach(), g

    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    launch, graph = "eager", None
    if not args.no_graph:
        try:
            state.clear()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(args.steps):
                    step()
            graph.replay()
            torch.cuda.synchronize()
            launch = "hipgraph"
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly\n")
            graph = None
            torch.cuda.synchronize()
            step()

    def region():
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                step()

    regions = timed_regions(region, args, world, dev)
    elapsed = regions[len(regions) // 2]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    step(ev)
    torch.cuda.synchronize()
    phases = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    gnorm = float(state["g"].norm())
    # algorithmic HBM bytes of the step per audio sample: render 8 (noise in, audio out); PQMF analysis
    # 8 (4 in + 4 out); each STFT resolution 4 in + 4 (n_fft/2+1)/hop target read; backward: the cotangent of the audio
    # written and read once per loss (4 x 4 x 2), the render backward reads noise + cotangent (8)
    bins = sum((n // 2 + 1) / h for n, h in ((1024, 120), (2048, 240), (512, 50)))
    bytes_per_sample = 8 + 8 + 3 * 4 + 4 * bins + 32 + 8
    algo = bytes_per_sample * B * T
    ms = elapsed / args.steps * 1e3
    pipes = gradstep_pipes(ms, B)
    return {
        "metric": "audio-seconds rendered+lossed+differentiated/sec, BASELINE configs[4] per-GPU share: 64-band PQMF + "
                  "3-resolution STFT loss, forward + backward to the 78 parameters",
        "value": round(world * B * SECONDS * args.steps / elapsed, 1), "unit": "audio-s/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "ms_per_step_min": round(regions[0] / args.steps * 1e3, 4), "timed_regions": len(regions),
        "timed_region_s": round(elapsed, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4] share: grad step, batch {B} x {SECONDS:g} s @ {SAMPLE_RATE} Hz per GPU, "
                               "PQMF(64) sub-band L1 + MR-STFT (1024/2048/512)", "batch_per_gpu": B, "launch": launch,
                   "loss": float(state["loss"]), "grad_norm": gnorm,
                   "phase_ms_eager": {"render_forward": round(phases[0], 4), "losses_forward": round(phases[1], 4),
                                      "backward": round(phases[2], 4)}},
        "roofline": {"kernel": "whole gradient step (no single dominant kernel: ~20 launches)", "bound": "hbm",
                     "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "pipes": pipes,
                     "algorithmic_bytes_per_step": int(algo), "bytes_per_sample": round(bytes_per_sample, 1)},
    }


def main():
    args = parse()
    if args.leg_child:
        return run_leg_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # IAS_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (the ranks share
    # the visible devices round-robin and the timing collectives run on CPU tensors); the real runs use RCCL.
    backend = os.environ.get("IAS_BENCH_BACKEND", "nccl")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))   # host-side setup only; no oversubscription
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(dev)
    global COLL_DEV
    COLL_DEV = dev if backend == "nccl" else torch.device("cpu")

    from inverse_audio_synthesis_amd import _lib
    from inverse_audio_synthesis_amd.pqmf import PQMF
    from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    _lib.load()

    if args.workload in ("vicreg", "gradstep"):
        result = (run_vicreg if args.workload == "vicreg" else run_gradstep)(args, rank, world, dev)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    B = args.batch or BATCH
    cfg = SynthConfig(batch_size=B, sample_rate=SAMPLE_RATE, buffer_size_seconds=SECONDS, reproducible=False)
    T = cfg.buffer_size
    voice = Voice(cfg).to(dev)
    gram = PQMF(N=3).to(dev)
    mel_l1 = MelSpectrogramL1(sample_rate=SAMPLE_RATE).to(dev)
    params = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000 + rank)).to(dev)
    tgt_params = torch.rand(B, 78, generator=torch.Generator().manual_seed(2000 + rank)).to(dev)
    target_mel = mel_l1.target(voice.render(tgt_params)).clone()
    voice.set_parameters01(params)

    ev = {"begin": [], "end": []}
    cev = {"pqmf": [], "stft": [], "control": []}      # (begin, end) HIP events on the stages' own streams (in-step durations)
    # instrument["on"]: False, "events" (HIP events around the stages: eager passes only) or "stamps" (ias_stamp launches
    # around the stages: device timestamps that can be captured INTO a graph, so the in-step durations of the line belong
    # to a replayed graph like the one the timed regions replay -- HIP events cannot be read back from a replay)
    instrument = {"on": False}
    STAGES = ("render", "pqmf", "stft", "control")
    stamp_buf = torch.zeros((len(STAGES), max(args.steps, args.warmup, 2), 2), dtype=torch.int64, device=dev)
    stamp_n = {n: 0 for n in STAGES}

    def stamp(name, which):
        i = stamp_n[name]
        cell = stamp_buf[STAGES.index(name), i, which:which + 1]
        _lib.check(_lib.load().ias_stamp(ctypes.c_void_p(cell.data_ptr()), _lib.stream()), "ias_stamp")
        if which == 1:
            stamp_n[name] = i + 1

    def bracket(name):
        """with bracket("pqmf"): ... -- timestamps on the CURRENT stream around the launches inside"""
        class _B:
            def __enter__(self_):
                if instrument["on"] == "events":
                    self_.e0 = torch.cuda.Event(enable_timing=True); self_.e0.record()
                elif instrument["on"] == "stamps":
                    stamp(name, 0)
            def __exit__(self_, *exc):
                if instrument["on"] == "events":
                    e1 = torch.cuda.Event(enable_timing=True); e1.record()
                    cev[name].append((self_.e0, e1))
                elif instrument["on"] == "stamps":
                    stamp(name, 1)
        return _B()

    def hook(name, phase):
        if name != "oscillators":
            return
        if instrument["on"] == "events":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev[phase].append(e)
        elif instrument["on"] == "stamps":
            stamp("render", 0 if phase == "begin" else 1)

    # Dataflow of one step: control-rate pass -> audio-rate render -> audio -> {PQMF, spectral loss}.  The two consumers
    # depend only on the audio, so they run on two side HIP streams (the control pass on a third); with several audio
    # buffers and workspaces in flight the NEXT steps' renders run beside this step's PQMF / STFT.  All K steps and their
    # cross-stream dependencies are captured once into one hipGraph and replayed (issue order matters: see run_steps).
    # IAS_BENCH_PRIO (diagnostics): "render" = the render's stream at high priority (the step's critical queue since the
    # PQMF and the STFT got shorter), "consumers" = PQMF / STFT at high priority
    prio = os.environ.get("IAS_BENCH_PRIO", "")
    if prio == "render":
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    cpr = -1 if prio == "consumers" else 0
    side_a, side_b, side_c = torch.cuda.Stream(priority=cpr), torch.cuda.Stream(priority=cpr), torch.cuda.Stream()
    # the small reduction of the STFT's loss partials gets a queue of its own: behind the STFT it lengthens the step's
    # critical queue, on the control stream it made the NEXT step's control pass (and with it the next render) wait for
    # the previous STFT (rocprofv3 timeline, profiles/r03b_trace_*: a 60 us bubble on the render queue per step)
    side_d = torch.cuda.Stream()
    red_stream = {"control": side_c, "own": side_d}.get(os.environ.get("IAS_BENCH_REDUCE_STREAM", "own"), side_d)
    nbuf = max(2, args.buffers)
    audio_bufs = [torch.empty((B, T), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    workspaces = [voice.new_workspace(dev) for _ in range(nbuf)]
    last = {}

    defer = not args.no_defer_consumers
    # IAS_BENCH_PRECLEAR=1: the render's 90 KB memset issued ahead on the control stream instead of in front of the render
    # (measured: 0.200 vs 0.183 ms per step -- the memset node elsewhere changes the executor's queue assignment for the
    # worse -- so it is off; kept as a knob)
    preclear = os.environ.get("IAS_BENCH_PRECLEAR", "0") == "1" and not args.no_pipeline
    reduce_aside = (not args.no_pipeline) and os.environ.get("IAS_BENCH_REDUCE_INLINE") != "1"   # (diag knob)
    # IAS_BENCH_EXTRA_FILL=n (diagnostics): n more 90 KB memset nodes in front of every render, like the one it starts with
    extra_fill = int(os.environ.get("IAS_BENCH_EXTRA_FILL", "0"))
    dummy_fill = torch.zeros(90 * 1024, dtype=torch.uint8, device=dev)

    warmed = {}

    def run_steps(k, pipelined=True):
        main = torch.cuda.current_stream()
        consumed = [None] * nbuf
        ws_free = [None] * nbuf     # audio pass that last read workspace[buf] has finished
        side_c.wait_stream(main)
        side_d.wait_stream(main)

        def issue_control(i):
            # the small control-rate kernels of step i go to their own stream with a private workspace,
            # so they run beside the previous step's audio-rate kernel instead of ahead of it
            buf = i % nbuf
            with torch.cuda.stream(side_c):
                if ws_free[buf] is not None:
                    side_c.wait_event(ws_free[buf])
                with bracket("control"):
                    voice.render_control(workspaces[buf])
                if preclear:
                    # the re-zeroing of the render's polled words leaves the render queue too (it was a memset node
                    # between every two renders): here it has to wait for the readers of the row peaks as well
                    if consumed[buf] is not None:
                        for e in consumed[buf]:
                            side_c.wait_event(e)
                    voice.clear_chain(workspaces[buf])
                return side_c.record_event()

        pending = [None]

        def issue_consumers(audio, peaks, rendered, buf):
            side_a.wait_event(rendered)
            side_b.wait_event(rendered)
            order = os.environ.get("IAS_BENCH_CONSUMERS", "parallel")   # diagnostics: how the two consumers are issued
            if order == "parallel":
                with torch.cuda.stream(side_a):
                    with bracket("pqmf"):
                        z = gram.analysis(audio.unsqueeze(1), rowpeak=peaks)
                    ea = side_a.record_event()
                with torch.cuda.stream(side_b):
                    # the STFT queue bounds the step: the 6 us reduction of its partials goes to the control stream
                    with bracket("stft"):
                        loss = mel_l1(audio, target_mel=target_mel, rowpeak=peaks,
                                      reduce_stream=red_stream if reduce_aside else None)
                    eb = side_b.record_event()
            else:
                with torch.cuda.stream(side_a):
                    if order == "pqmf_first":
                        z = gram.analysis(audio.unsqueeze(1), rowpeak=peaks)
                        loss = mel_l1(audio, target_mel=target_mel, rowpeak=peaks)
                    else:
                        loss = mel_l1(audio, target_mel=target_mel, rowpeak=peaks)
                        z = gram.analysis(audio.unsqueeze(1), rowpeak=peaks)
                    ea = eb = side_a.record_event()
            consumed[buf] = (ea, eb)
            last["z"], last["loss"] = z, loss

        # control passes run `depth` steps ahead of their render
        # (two steps of lead, legal with three workspaces, measured slower: 0.232 vs 0.211 ms -- the capture order of the
        # extra successor changes the executor's queue assignment again)
        depth = 1
        # IAS_BENCH_NOCTRL=1 (diagnostics, scripts/diag/run_noctrl_ab.sh): the control pass only for the first `nbuf` steps
        # (the bench's parameters do not change, so every workspace keeps valid control signals) -- what the step would
        # cost if the control pass were free
        noctrl = os.environ.get("IAS_BENCH_NOCTRL", "0") == "1"
        real_issue_control = issue_control
        if noctrl:
            def issue_control(i):  # noqa: F811
                return real_issue_control(i) if not warmed.get("done") else side_c.record_event()
        ctrl_events = {j: issue_control(j) for j in range(min(depth, k))}
        for i in range(k):
            buf = i % nbuf
            if consumed[buf] is not None:          # buffer free again: both readers of step i-nbuf are done
                for e in consumed[buf]:
                    main.wait_event(e)
            main.wait_event(ctrl_events.pop(i))
            early_ctrl = pipelined    # the control pass of the next step is issued ahead of this step's render
            if early_ctrl and i + depth < k:
                ctrl_events[i + depth] = issue_control(i + depth)
            # normalize_if_clipping is folded into the two consumers (row peaks read in place from the workspace)
            for _ in range(extra_fill):            # (diagnostics: is the render queue the step's critical chain?)
                dummy_fill.zero_()
            audio = voice.render_audio(workspaces[buf], out=audio_bufs[buf], on_stage=hook, normalize=False,
                                       precleared=preclear)
            peaks = voice.peaks_view(workspaces[buf])
            ws_free[buf] = main.record_event()
            if not early_ctrl and i + depth < k:
                ctrl_events[i + depth] = issue_control(i + depth)
            step_out = (audio, peaks, ws_free[buf], buf)
            if defer and pipelined:
                # Issue order = capture order.  The graph executor keeps the FIRST successor of a node on that node's
                # hardware queue: with the consumers issued right after their render, the PQMF took the render's queue
                # and the next render waited behind it (rocprofv3 trace: render(i) -> pqmf(i) -> render(i+1) on one queue,
                # a 56 us bubble per step).  Issuing the consumers of step i-1 after render(i) makes the next render the
                # first successor; the dependencies are unchanged.
                if pending[0] is not None:
                    issue_consumers(*pending[0])
                pending[0] = step_out
            else:
                issue_consumers(*step_out)
                if not pipelined:
                    for e in consumed[buf]:
                        main.wait_event(e)
        if pending[0] is not None:
            issue_consumers(*pending[0])
            pending[0] = None
        main.wait_stream(side_a)
        main.wait_stream(side_b)
        main.wait_stream(side_c)
        main.wait_stream(side_d)
        warmed["done"] = k >= nbuf

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    pipelined = not args.no_pipeline
    # ---- warm-up (eager), optional hipGraph capture of the K-step schedule
    run_steps(max(args.warmup, 2), pipelined)
    torch.cuda.synchronize()
    launch = "eager"
    graph = None
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                run_steps(args.steps, pipelined)
            graph.replay()
            torch.cuda.synchronize()
            launch = "hipgraph"
        except Exception as e:  # noqa: BLE001 -- report and fall back to eager launches
            sys.stderr.write(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly\n")
            graph = None
            torch.cuda.synchronize()

    # ---- timed regions: each one is EXACTLY K steps between barrier+synchronize brackets; R of them are timed
    # one after the other (a single 5 ms region is dominated by pipeline fill/drain and clock ramp) and the MEDIAN
    # region gives `value` / `ms_per_step`; the fastest one is reported next to it.
    def one_region():
        sync_all()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            run_steps(args.steps, pipelined)
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=COLL_DEV)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el

    first = one_region()
    replays = args.replays if args.replays > 0 else max(20, min(2000, int(0.25 / max(first, 1e-6)) + 1))
    if world > 1:   # every rank must time the same number of regions
        r = torch.tensor([replays], dtype=torch.int64, device=COLL_DEV)
        dist.broadcast(r, 0)
        replays = int(r.item())
    regions = sorted(one_region() for _ in range(replays))
    elapsed = regions[len(regions) // 2]
    elapsed_min = regions[0]
    # the look-back chains of every render in the timed regions completed (a non-zero status word = NaN audio)
    for w_ in workspaces:
        assert voice.chain_status(w_) == 0, "voice render: a tile's bounded wait expired during the benchmark"
    loss_value = last["loss"].item()

    # ---- dominant-kernel timing with HIP events on the launch stream (eager: events cannot be read back
    # from inside a replayed graph).  Two passes over the same K steps:
    #   isolated   -- the kernel's own duration (below), the one the roofline is computed from (agrees with
    #                 rocprofv3 of `bench.py --no-pipeline`, profiles/*_nopipeline_kernel_stats.csv);
    #   overlapped -- the pipelined schedule of the timed region, where the kernel shares the GPU with the
    #                 previous step's PQMF / STFT and the next step's control pass (agrees with rocprofv3 of
    #                 the default command); longer per launch, shorter per step.
    # The bracket holds the kernel and the 90 KB memset of its ticket / aggregate words.
    def timed_pass(pipe):
        ev["begin"].clear(); ev["end"].clear()
        cev["pqmf"].clear(); cev["stft"].clear(); cev["control"].clear()
        instrument["on"] = "events"
        torch.cuda.synchronize()
        run_steps(args.steps, pipe)
        torch.cuda.synchronize()
        instrument["on"] = False
        ms = [b.elapsed_time(e) for b, e in zip(ev["begin"], ev["end"])]
        out = {"render": sum(ms) / len(ms)}
        for name in ("pqmf", "stft"):
            ms_ = [b.elapsed_time(e) for b, e in cev[name]]
            out[name] = sum(ms_) / len(ms_) if ms_ else None
        return out

    def stamped_replay(pipe):
        """The K-step schedule captured ONCE MORE with a device timestamp before and after every stage (six one-thread
        launches per step on the stages' own streams) and replayed: -> in-step duration per stage [ms], step time of the
        instrumented replay [ms] (next to ms_per_step it shows what the stamps cost)."""
        for n in STAGES:
            stamp_n[n] = 0
        stamp_buf.zero_()
        instrument["on"] = "stamps"
        torch.cuda.synchronize()
        gi = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gi):
            run_steps(args.steps, pipe)
        instrument["on"] = False
        gi.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gi.replay(); e1.record()
        torch.cuda.synchronize()
        t = stamp_buf.cpu().double()[:, :args.steps]
        if os.environ.get("IAS_BENCH_DUMP_STAMPS"):            # diagnostics: the raw timeline of the instrumented replay [us]
            t0_ = t[t > 0].min()
            with open(os.environ["IAS_BENCH_DUMP_STAMPS"], "w") as f:
                for i_ in range(args.steps):
                    f.write("step %2d  " % i_ + "  ".join("%s %8.1f -> %8.1f" % (n, (t[k_, i_, 0] - t0_) * 1e-2, (t[k_, i_, 1] - t0_) * 1e-2)
                                                          for k_, n in enumerate(STAGES)) + "\n")
        dur = (t[:, :, 1] - t[:, :, 0]) * 1e-5                # 100 MHz ticks -> ms
        return {n: float(dur[i][dur[i] > 0].mean()) if (dur[i] > 0).any() else None for i, n in enumerate(STAGES)}, e0.elapsed_time(e1) / args.steps

    instep_source, stamped_step_ms = "HIP events around the stages in an eager pass of the schedule", None
    instep_ms = None
    if graph is not None:
        try:
            instep_ms, stamped_step_ms = stamped_replay(pipelined)
            instep_source = ("device timestamps (ias_stamp: s_memrealtime, 10 ns) captured into a second graph of the same "
                             "K-step schedule, one replay; includes the launch gap behind each stamp")
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench] stamped replay failed ({type(e).__name__}: {e}); eager events instead\n")
            instrument["on"] = False
            torch.cuda.synchronize()
    if instep_ms is None:
        instep_ms = timed_pass(pipelined)

    def isolated_ms(fn):
        """K back-to-back launches of one stage captured in a graph (no host launch gap inside), HIP events around the replay"""
        fn(); fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk):
                for _ in range(args.steps):
                    fn()
            gk.replay()
            torch.cuda.synchronize()
            a0.record(); gk.replay(); a1.record()
        except Exception:  # noqa: BLE001
            torch.cuda.synchronize()
            a0.record()
            for _ in range(args.steps):
                fn()
            a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / args.steps

    # isolated: K back-to-back launches of the audio-rate stage alone (its 90 KB memset + the kernel, control
    # signals already in the workspace), captured in a graph so that no host launch gap falls inside the
    # bracket, timed with HIP events around the replay
    def osc_only(k):
        for _ in range(k):
            voice.render_audio(workspaces[0], out=audio_bufs[0], normalize=False)

    voice.render_control(workspaces[0])
    osc_only(2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            osc_only(args.steps)
        g2.replay()
        torch.cuda.synchronize()
        e0.record(); g2.replay(); e1.record()
    except Exception:  # noqa: BLE001
        torch.cuda.synchronize()
        e0.record(); osc_only(args.steps); e1.record()
    torch.cuda.synchronize()
    osc_ms_avg = e0.elapsed_time(e1) / args.steps

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * SECONDS * args.steps / elapsed

    # ---- the three audio-rate kernels of the step, each alone (isolated) and inside the pipelined schedule (in-step):
    # algorithmic HBM bytes per sample: render 8 (noise in + audio out), PQMF 8 (in + out),
    # mel-L1 against a cached target 4 + 4 * n_mels / hop = 5
    peaks0 = voice.peaks_view(workspaces[0])
    plan = mel_l1.mel.plan
    n_part = _lib.load().ias_stft_partials_count(B, T, plan.n_fft, plan.hop_length,
                                                (0 if plan.mtables is None else 1) | 2 | (0 if plan.segtab is None else 4))
    parts = torch.empty((n_part, 3), dtype=torch.float64, device=dev)
    from inverse_audio_synthesis_amd.spectral import LOSS_L1, VALUE_POWER
    iso_ms = {
        "render": osc_ms_avg,
        "pqmf": isolated_ms(lambda: gram.analysis(audio_bufs[0].unsqueeze(1), rowpeak=peaks0)),
        "stft": isolated_ms(lambda: plan._call(audio_bufs[0], None, target_mel, parts, VALUE_POWER, LOSS_L1, 0.0, peaks0)),
    }
    ctrl_iso_ms = isolated_ms(lambda: voice.render_control(workspaces[0]))
    bps = {"render": RENDER_BYTES_PER_SAMPLE, "pqmf": 8.0, "stft": 4.0 + 4.0 * plan.n_out / plan.hop_length}
    knames = {"render": "voice_audio_kernel",
              "pqmf": "pqmf_analysis_mods_kernel",
              "stft": "stft2_kernel<8, true, 1, 1>" if plan.n_fft == 1024 else "stft_kernel"}
    table = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            table = json.load(open(tpath))      # keys are rocprofv3 kernel names: "void voice_audio_kernel<0, true>"
        except Exception:  # noqa: BLE001
            table = {}

    def pmc_traffic(kname):
        hits = [v for k, v in table.items() if kname in k and isinstance(v, dict)]
        return hits[0].get("hbm_bytes_per_launch") if hits else None

    # SQ counters per launch (profiles/counters.json: scripts/make_counters.py from the rocprofv3 --pmc summaries) and the
    # static VALU cost per instruction (profiles/isa_costs.json: scripts/isa_costs.py on the ISA of the built kernels):
    # imported like `traffic`, recomputable from profiles/<round>_pmc_{voice,stft,pqmf}.txt
    def load_json(name):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:  # noqa: BLE001
            return {}
    counters, isa_costs = load_json("counters.json"), load_json("isa_costs.json")
    # the counters belong to the build they were profiled on: scripts/make_counters.py stores a hash of the three kernels'
    # sources and flags; when the sources have moved on, the per-pipe fractions are marked stale and no limiter is named
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from make_counters import kernel_sources_sha16
        counters_stale = counters.get("_source_sha16") != kernel_sources_sha16(ROOT)
    except Exception:  # noqa: BLE001
        counters_stale = True
    N_CU = torch.cuda.get_device_properties(dev).multi_processor_count      # 256 on MI355X
    N_SIMD, CLOCK_MHZ = 4 * N_CU, 2100.0     # shader clock under these kernels: 2.10-2.15 GHz (s_memtime / s_memrealtime)
    # SQ_INSTS_VALU counts wave instructions: the unit is what ONE WAVE INSTRUCTION STREAM covers
    units = {"render": ("sample per lane (a wave: 64 samples)", B * T / 64.0),
             "pqmf": ("4 frames per lane (a wave: 256 frames)", B * (T // 3) / 256.0),
             "stft": ("frame (one wave per frame)", B * (1 + T // plan.hop_length))}

    kernels = {}
    for name in ("render", "pqmf", "stft"):
        ab = bps[name] * B * T
        gbs = ab / (iso_ms[name] * 1e-3) / 1e9
        k = {"kernel": knames[name], "algorithmic_bytes_per_launch": int(ab),
             "isolated_avg_us": round(iso_ms[name] * 1e3, 1),
             "in_step_avg_us": None if instep_ms[name] is None else round(instep_ms[name] * 1e3, 1),
             "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
             "traffic": pmc_traffic(knames[name])}
        c = next((v for kk, v in counters.items() if isinstance(v, dict) and knames[name].split("<")[0] in kk), None)
        cost = next((v for kk, v in isa_costs.items() if isinstance(v, dict) and knames[name].split("<")[0] in kk), None)
        cycles = iso_ms[name] * 1e-3 * CLOCK_MHZ * 1e6            # shader clocks of one isolated launch
        fracs = {"hbm": k["frac"]}
        if c and "SQ_INSTS_VALU" in c:
            k["valu_insts"] = int(c["SQ_INSTS_VALU"])
            k["valu_insts_per_unit"] = round(c["SQ_INSTS_VALU"] / units[name][1], 1)
            k["unit_of_work"] = units[name][0]
            if cost:
                k["valu_clk_per_inst"] = cost["valu_clk_per_inst"]
                k["frac_valu"] = round(c["SQ_INSTS_VALU"] * cost["valu_clk_per_inst"] / (N_SIMD * cycles), 4)
                fracs["valu"] = k["frac_valu"]
            if "SQ_LDS_IDX_ACTIVE" in c:
                k["lds_busy_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / (N_CU * cycles), 4)
                fracs["lds"] = k["lds_busy_frac"]
                if c["SQ_LDS_IDX_ACTIVE"] > 0 and "SQ_LDS_BANK_CONFLICT" in c:
                    k["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
            k["counters_round"] = counters.get("_round")
            k["counters_stale"] = counters_stale
        # what the counters say binds the kernel: the largest of its HBM, vector-pipe and LDS-pipe fractions
        k["bound"] = max(fracs, key=fracs.get) if not counters_stale else "hbm (counters stale: profiled on other sources)"
        kernels[name] = k
    # What the step's pipes are busy for (verdict r04 item 1a): per pipe, the sum over the three kernels of (busy fraction x
    # isolated time).  The kernels are work-conserving neighbours (the step equals the sum of their isolated times), so the
    # busiest pipe's sum is the floor this set of kernels can reach if nothing ever waited: the attainable step, and the
    # chain's HBM fraction at it, next to the 0.70 the north_star asks for.
    pipe_busy_us = {"valu": 0.0, "lds": 0.0, "hbm": 0.0}
    for k_ in kernels.values():
        iso_us = k_["isolated_avg_us"]
        pipe_busy_us["valu"] += k_.get("frac_valu", 0.0) * iso_us
        pipe_busy_us["lds"] += k_.get("lds_busy_frac", 0.0) * iso_us
        pipe_busy_us["hbm"] += k_["frac"] * iso_us
    have_counters = all("frac_valu" in k_ for k_ in kernels.values())
    floor_pipe = max(pipe_busy_us, key=pipe_busy_us.get)
    pipe_floor_ms = pipe_busy_us[floor_pipe] * 1e-3 if have_counters else None
    # the dominant kernel of the TIMED schedule = the one with the largest in-step duration per step
    dom = max(kernels, key=lambda n: kernels[n]["in_step_avg_us"] or 0.0)
    chain_bytes = sum(bps.values()) * B * T          # 21 B here; SURVEY.md rounds the chain to 22 B/sample (497 MB)
    chain_frac = 22.0 * B * T / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
    algo_bytes = kernels[dom]["algorithmic_bytes_per_launch"]
    achieved = kernels[dom]["achieved_GBps"]
    traffic = kernels[dom]["traffic"]
    kdesc = {"render": "voice_audio_kernel (phase increments + chained fp64 scan + oscillators + mixer)",
             "pqmf": knames["pqmf"] + " (PQMF(3) analysis" + (" in its cosine-modulated form)" if "mod" in knames["pqmf"]
                                                                 else " on v_mfma_f32_16x16x4_f32)"),
             "stft": knames["stft"].split("<")[0] + " (framed radix-8 FFT + power + mel projection + L1 against the cached target)"}

    result = {
        "metric": "audio-seconds rendered+lossed/sec (whole node), batch=128 4s@44.1kHz",
        "value": round(value, 1),
        "unit": "audio-s/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_min": round(elapsed_min / args.steps * 1e3, 4),
        "timed_regions": replays, "timed_region_s": round(elapsed, 5), "timed_total_s": round(sum(regions), 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: torchsynth-style Voice render + PQMF(3) analysis + mel-L1 loss, "
                        f"batch {B} x {SECONDS:g} s @ {SAMPLE_RATE} Hz per GPU",
            "batch_per_gpu": B, "samples_per_voice": T, "launch": launch, "streams": 5, "pipelined": pipelined, "loss": loss_value,
        },
        "roofline": {
            "kernel": kdesc[dom],
            "dominant_by": "largest in-step duration per step of the timed (pipelined) schedule",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "limited_by": kernels[dom]["bound"],
            "limited_by_note": "`bound`/`frac` price the kernel against the HBM roofline as the contract asks; `limited_by` and "
                               "kernels.*.bound name the pipe the SQ counters show busiest (frac = HBM, frac_valu = SQ_INSTS_VALU x "
                               "static clocks per instruction / (1024 SIMDs x clocks of an isolated launch at 2.1 GHz), lds_busy_frac = "
                               "SQ_LDS_IDX_ACTIVE / (256 CUs x the same clocks); counters imported from profiles/counters.json, "
                               "instruction costs from profiles/isa_costs.json)",
            "traffic_source": "imported from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the "
                              "kernels, gfx950 x2 read correction); not measured by this run",
            "avg_launch_ms": round(iso_ms[dom], 4), "algorithmic_bytes_per_launch": algo_bytes,
            "measured": "HIP events around K back-to-back launches of the stage captured in a graph, on the stream it is "
                        "launched on (isolated); in_step: " + instep_source,
            "overlapped_avg_launch_ms": None if instep_ms[dom] is None else round(instep_ms[dom], 4),
            "stamped_replay_ms_per_step": None if stamped_step_ms is None else round(stamped_step_ms, 4),
            "kernels": kernels,
            "chain_bytes_per_step": int(22.0 * B * T), "chain_bytes_listed_kernels": int(chain_bytes),
            "chain_frac": round(chain_frac, 4),
            "pipe_busy_us_per_step": {k_: round(v_, 1) for k_, v_ in pipe_busy_us.items()},
            "pipe_floor_ms": None if pipe_floor_ms is None else round(pipe_floor_ms, 4),
            "pipe_floor_pipe": floor_pipe if have_counters else None,
            "chain_frac_at_floor": None if pipe_floor_ms is None else round(22.0 * B * T / (pipe_floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "step_over_floor": None if pipe_floor_ms is None else round(ms_per_step / pipe_floor_ms, 3),
            "pipe_floor_note": "per pipe: sum over render / PQMF / STFT of (busy fraction x isolated time) -- frac_valu, lds_busy_frac, "
                               "HBM frac; the largest sum is the step these kernels could reach with perfect overlap and no waiting "
                               "(the control pass and launch gaps excluded); chain_frac_at_floor = the chain's 22 B/sample over it",
            "control_pass_isolated_us": round(ctrl_iso_ms * 1e3, 1),
            "control_pass_in_step_us": None if not instep_ms.get("control") else round(instep_ms["control"] * 1e3, 1),
            "chain_frac_note": "22 B/sample (SURVEY.md 8d: render 8 + PQMF 8 + mel-L1 5, rounded up) x B x T / ms_per_step "
                               "/ 8 TB/s: the whole step against the HBM roofline (north_star target 0.70)",
        },
    }
    # ---- driver-timed secondary legs (N = 1 only, after the headline's timed regions, never inside them): the MFMA
    # Gram of configs[2] / [3]'s loss and the configs[4] gradient step, a few hundred ms each
    if rank == 0 and world == 1 and not args.no_legs:
        result["legs"] = secondary_legs(args, dev)
    if world > 1 and not args.no_legs:
        # the legs with a real exchange step, on every rank (children on a process group of their own: multi_gpu_legs)
        legs = multi_gpu_legs(args, rank, world, local_rank)
        if rank == 0:
            result["legs"] = legs
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only (bench contract)
            result["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
