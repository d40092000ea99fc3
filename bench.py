"""Benchmark of the hot path: 10 s @ 16 kHz clips/s through encoder + frame head + per-frame decode.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): wav2vec2-base, 32 x 10 s clips per GPU, bf16 MFMA operands,
synthetic waveform (seed 1986, 0.1*randn clamped to [-1,1]) already resident in HBM, seeded random weights.
One step = encoder forward (incl. both whole-batch layer norms) -> 20-way head -> sigmoid/argmax frame
decode kernel (+ one all-gather of the logits over RCCL when N > 1).  Weak scaling: the per-GPU batch is
fixed, ranks own disjoint clips.  Prints ONE JSON line on rank 0.

Successive steps are issued round-robin on two HIP streams (--streams, each with its own encoder object and
workspace): every step still computes its whole batch, but the HBM-bound kernels of one step overlap the MFMA-bound
kernels of the next.  The roofline leg (HIP events around every launch of the dominant kernel) replays the K steps on
one stream after the timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# MI355X dense peaks (MI355X_MICROARCH.md).  The split-operand modes issue three 16-bit MFMAs per algorithmic
# multiply-add (Ah*Wh + Al*Wh + Ah*Wl), so their ceiling in ALGORITHMIC flops is a third of the bf16 / fp16 peak.
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3, "fp16x3": 2500.0 / 3}


def synth_wav(B, L, seed=1986):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)


def cpu_baseline(cfg, sd, hd, seconds, budget_s=25.0):
    """The oracle (CPU fp32 restatement of the reference forward, kind="port") on the host cores.
    ATen's intra-op threading stops scaling well before 128 cores on this workload (measured on the GPU box:
    16 threads 3.95 clips/s, 64 threads 2.5, 128 threads 1.1), so a short sweep picks the best thread count
    and `cores` reports the threads actually used for the quoted number."""
    from oracle import svt_oracle as O
    ncpu = os.cpu_count() or 1
    B = 4
    wav = synth_wav(B, int(16000 * seconds), seed=1986)

    def one():
        with torch.no_grad():
            f = O.encoder_forward(sd, cfg, wav)
            lg = O.head_forward(f, hd["w.weight"], hd["w.bias"])
            O.decode_frames(lg)

    t_start = time.perf_counter()
    best = None
    tried = []
    for nt in [t for t in (16, 32, 64) if t <= ncpu] or [ncpu]:
        if time.perf_counter() - t_start > budget_s:
            break
        torch.set_num_threads(nt)
        one()  # warm-up at this thread count
        t = time.perf_counter()
        one()
        dt = time.perf_counter() - t
        tried.append((nt, round(B / dt, 3)))
        if best is None or dt < best[1]:
            best = (nt, dt)
    nt, dt = best
    return {"value": round(B / dt, 4), "unit": "clips/s", "cores": nt, "kind": "port",
            "sample": f"oracle/svt_oracle.py fp32 (torch CPU), {B} x {seconds:g} s clips per pass, 1 warm-up + 1 timed pass "
                      f"per thread count, best of {tried} (threads, clips/s); host has {ncpu} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="wav2vec2-base")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3", "fp16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h2d", action="store_true", help="diagnostic: every step first copies its batch from pinned host memory "
                    "(the PCIe-inclusive rate; never the reported metric, whose inputs are resident in HBM)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (measured: no gain, the GPU is never idle)")
    ap.add_argument("--streams", type=int, default=2, help="issue successive steps round-robin on this many HIP streams (each with its own encoder "
                    "object and workspace), so one step's HBM-bound kernels (LayerNorm, conv0, norms) run under the next step's MFMA-bound "
                    "ones: measured +5 %% with 2, less with 3.  The roofline leg always runs on one stream.")
    args = ap.parse_args()

    import svt_speechbrain_amd as S
    from svt_speechbrain_amd import _lib, distributed as D
    from svt_speechbrain_amd import weights as W

    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))  # before the process group: RCCL binds to it
    rank, local, world = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device(f"cuda:{local}")
    lib = _lib.load()
    _lib.require_gpu()
    for kv in os.environ.get("SVT_DEBUG_SET", "").split(","):  # diagnostics: "key=value,..." -> svt_debug_set (A/B of kernel variants)
        if "=" in kv:
            lib.svt_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))

    cfg = S.PRESETS[args.model]
    L = int(16000 * args.seconds)
    T = cfg.frames(L)
    B = args.batch
    n_total = B * world
    lo, hi = D.shard_bounds(n_total, rank, world)
    # rank r owns clips [lo, hi) of the global batch; each clip is seeded by its global index, so no rank ever
    # materialises another rank's clips and the union over ranks is the same for every N
    if world > 1:
        wav = torch.cat([synth_wav(1, L, seed=1986 + i) for i in range(lo, hi)]).to(dev)
    else:
        wav = synth_wav(B, L).to(dev)

    ns = max(1, args.streams)
    enc = S.HuggingFaceWav2Vec2(args.model, None, config=cfg, precision=args.precision, seed=1986).to(dev)
    encs = [enc] + [enc.replica() for _ in range(ns - 1)]  # same parameters, own device handle + workspace per stream
    head = S.Linear(20, input_size=cfg.hidden_size)
    hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=2986)
    head.load_state_dict(hd)
    head = head.to(dev)
    frames_l = [torch.empty((B * T, 4), dtype=torch.int32, device=dev) for _ in range(ns)]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(ns - 1)]
    wav_host = wav.cpu().pin_memory() if args.h2d else None
    wav_in = [torch.empty_like(wav) for _ in range(ns)] if args.h2d else None
    counter = [0]
    active = [ns]  # streams in use (the roofline leg sets this to 1)

    import contextlib

    def step():
        i = counter[0] % active[0]
        counter[0] += 1
        # one stream: launch on whatever stream is current (during hipGraph capture that is the capture stream)
        with (torch.cuda.stream(streams[i]) if ns > 1 else contextlib.nullcontext()):
            if args.h2d:
                wav_in[i].copy_(wav_host, non_blocking=True)
            feats = encs[i](wav_in[i] if args.h2d else wav)
            logits = head(feats)
            _lib.check(lib.svt_decode_frames(_lib.ptr(logits), B * T, 20, 4, 12, _lib.ptr(frames_l[i]), local,
                                             _lib.stream_ptr(dev)), "svt_decode_frames")
            if world > 1:
                return D.all_gather_rows(logits, n_total, world)
        return logits

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    # The step is ~110 dependent kernel launches with no host decisions in between: capture it once into a hipGraph
    # (the C-ABI forward neither allocates nor synchronises) and replay it; every replay executes the full step.
    graph = None
    if args.graph and world == 1 and ns == 1:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as ex:  # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture unavailable ({type(ex).__name__}: {ex}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    # HIP-event pairs around every SAMPLE-th dense-contraction launch of the timed region itself (on the launch stream): the
    # live measurement.  Sampling keeps the cost of the event records (~5 % of a step when every launch carries a pair) below 1 %.
    SAMPLE = 8
    lib.svt_prof_reset()
    if graph is None:
        lib.svt_prof_enable(SAMPLE)
    D.barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            out = step()
    torch.cuda.synchronize()
    D.barrier(world)
    elapsed = time.perf_counter() - t0
    lib.svt_prof_enable(0)
    elapsed = D.max_over_ranks(elapsed, world, dev)
    assert out.shape[0] == n_total
    # a step is >= ~110 kernels: anything faster than this did not run the work (e.g. an empty captured graph)
    if 1e3 * elapsed / args.steps < 0.05:
        raise SystemExit("[bench] implausible step time: the timed region did not execute the step")

    def prof(kind):
        n, ms_, fl_, by_ = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        _lib.check(lib.svt_prof_read(kind, C.byref(n), C.byref(ms_), C.byref(fl_), C.byref(by_)), "svt_prof_read")
        return n.value, ms_.value, fl_.value, by_.value

    live = prof(0)

    # roofline leg: the SAME K steps again with a HIP-event pair around every launch of the dense-contraction kernels
    # (on the stream they are launched on).  Kept out of the throughput timing above because 2 event records per
    # launch x ~70 launches per step add ~5 % of GPU idle time.
    # It runs on ONE stream: with two, kernels of consecutive steps share the chip and a launch lasts ~1.2x longer while the
    # job finishes sooner -- a per-launch duration under overlap says nothing about the kernel.
    active[0] = 1
    counter[0] = 0
    lib.svt_prof_reset()
    lib.svt_prof_enable(1)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    lib.svt_prof_enable(0)

    k_dom, k_other, k_attn = prof(0), prof(1), prof(2)

    if rank == 0:
        clips_per_s = n_total * args.steps / elapsed
        peak = MFMA_PEAK_TFLOPS[args.precision]
        n_l, ms, fl, _ = k_dom
        achieved = (fl / 1e12) / (ms / 1e3) if ms > 0 else 0.0
        flops_clip = cfg.flops_per_clip(L)
        # HBM bytes per launch of the dominant kernel family from the committed PMC passes (separate rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 read correction applied: tools/pmc_summary.py);
        # null when that file is absent or the workload is not the default one
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
        if os.path.exists(pmc_file) and args.model == "wav2vec2-base" and B == 32 and args.seconds == 10.0 and args.precision == "bf16":
            try:
                pm = json.load(open(pmc_file))
                fam = [v for k, v in pm.items() if k.startswith(("gemm_pers_kernel", "gemm_pp8_kernel", "outproj_ln_kernel"))]
                tot_n = sum(v["launches"] for v in fam)
                traffic = round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in fam) / tot_n / 1e9, 4)
            except Exception:
                traffic = None
        res = {
            "metric": "10s@16kHz clips/sec encoder+CTC forward, wav2vec2-base, 1/2/4/8 MI355X",
            "value": round(clips_per_s, 3),
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"{args.model} audio-only AMT forward (encoder + 20-way head + frame decode), "
                                   f"{B} x {args.seconds:g} s @16 kHz mono clips per GPU",
                       "global_batch": n_total, "per_gpu_batch": B, "samples_per_clip": L, "frames_per_clip": T,
                       "gflop_per_clip": round(flops_clip / 1e9, 2), "parallelism": f"clips sharded over {world} rank(s)",
                       "launch": "hipGraph replay" if graph is not None else "eager",
                       "streams": ns, "inputs": "pinned host memory, copied every step (diagnostic)" if args.h2d else "resident in HBM",
                       "end_to_end_mfma_frac": round(clips_per_s / world * flops_clip / (peak * 1e12), 4)},
            # dominant kernel = svt::gemm_pp8_kernel<BM> (conv1-6, projection, q/k/v/out, FFN): algorithmic flops of its
            # launches / HIP-event time of those launches on their stream, over the timed region
            "roofline": {"bound": "mfma", "kernel": "svt::gemm_pers_kernel / gemm_pp8_kernel <BM=128|192|256> / outproj_ln_kernel (one LDS-DMA MFMA pipeline: persistent, one tile per workgroup, or row-complete with fused LayerNorm)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_unit": "GB of HBM traffic per launch (PMC, profiles/r01_pmc_hbm_traffic.json)",
                         "algorithmic_gb_per_launch": round(k_dom[3] / max(1, n_l) / 1e9, 4),
                         "launches": int(n_l), "avg_launch_ms": round(ms / max(1, n_l), 5),
                         "ms_per_step": round(ms / args.steps, 4),
                         "flops_per_launch_avg": round(fl / max(1, n_l), 1),
                         "note": "achieved / avg_launch_ms: HIP events around every launch, the K steps replayed on ONE stream right after "
                                 f"the timed region; the timed region itself issues steps round-robin on {ns} stream(s), where kernels of "
                                 "consecutive steps share the chip and a launch lasts longer while the job finishes sooner "
                                 "(in_timed_region: every 8th launch of the timed region, same events, same stream as the launch)",
                         "in_timed_region": {"launches_sampled": int(live[0]),
                                             "avg_launch_ms": round(live[1] / max(1, live[0]), 5),
                                             "achieved": round((live[2] / 1e12) / (live[1] / 1e3), 2) if live[1] > 0 else None,
                                             "streams": ns},
                         "other_kernels_ms_per_step": {"small/fp32 gemm": round(k_other[1] / args.steps, 4),
                                                       "flash_attn": round(k_attn[1] / args.steps, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            sd = {k[len("model."):]: v.detach().cpu() for k, v in enc.state_dict().items()}
            res["cpu_baseline"] = cpu_baseline(cfg, sd, hd, args.seconds)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
