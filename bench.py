#!/usr/bin/env python3
"""Headline benchmark: SP two-stream training step (BASELINE.json configs[1]/[2]).

One "step" = one pass of the hot path over one synthetic minibatch of 32 frames per GPU, exactly the body of
SP.trainSP's loop (reference SP.py:132-138): model_SP forward (train-mode BN) -> floss -> backward -> Adam ->
zero_grad, every kernel hand-written HIP behind the C-ABI.  Inputs are resident in HBM before the timed region.
N>1: one process per GPU (torch.distributed.run), gradient all-reduce over RCCL overlapped with backward.

Prints ONE JSON line on rank 0 (contract in the task prompt) with `roofline` (dominant kernel: the f32-MFMA
implicit-GEMM 3x3 conv, timed live with HIP events on the launch stream) and `cpu_baseline` (the oracle, i.e.
the reference's PyTorch-CPU algorithm, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md:41
F16_MFMA_PEAK_TFLOPS = 2500.0       # dense f16 / bf16 MFMA, :42
HBM_PEAK_GBS = 8000.0               # :35
FLOP_PER_FRAME_FWD_BWD = 341.2e9    # BASELINE.md section 3 (all parameters trainable)
BYTES_PER_FRAME = 1.046e9


def kernel_src_sha():
    """Short hash of the split-half conv kernel sources (what the committed PMC traffic profile is valid for)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("conv3x3_igemm_x3s.hip", "conv3x3_igemm_x3.hip", "x3_split.h", "conv3x3_wgrad.hip"):
        h.update(open(os.path.join(ROOT, "egocentric-gaze-prediction_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


class stdout_to_stderr:
    """RCCL prints a version banner to C stdout when its first communicator comes up; the contract of this script is ONE JSON
    line on stdout, so descriptor 1 points at stderr while a process group is initialised / first used.  The banner goes through
    C stdio, which is block-buffered when stdout is a pipe or a file: it has to be flushed BEFORE descriptor 1 is restored, or it
    surfaces at process exit, after the JSON line."""

    @staticmethod
    def _flush_c():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    def __enter__(self):
        sys.stdout.flush()
        self._flush_c()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def mfma_power_probe(dev, ms_target=30.0):
    """Sustained v_mfma_f32_32x32x16_f16 rate of THIS chip on random operand bits (csrc/probe.hip): register operands, no
    memory traffic.  MI355X manages its matrix cores against a power budget -- ~1.5 PFLOP/s on random bits vs 2.4 on zeros
    (profiles/r02_mfma_power_probe.txt) and the figure differs by several per cent from chip to chip -- so the bench line
    carries the ceiling of the box it ran on: roofline.power_ceiling_tflops (MFMA TFLOP/s; one algorithmic MAC of the
    split-half kernels costs 3 MFMA MACs) and the effective clock it implies (1024 flop / cycle / SIMD, 1024 SIMDs)."""
    import egaze_amd.hipops as H
    frag = torch.randn(16 * 64 * 8, device=dev).to(torch.float16)
    blocks = 2048
    out = torch.empty(blocks * 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.check(H._RAW_LIB.egz_mfma_probe(frag.data_ptr(), out.data_ptr(), blocks, iters, st), "egz_mfma_probe")
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    run(200)
    t = run(1000)
    iters = max(1000, int(1000 * ms_target / max(t, 1e-3)))
    ms = run(iters)
    tf = blocks * 4 * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12
    return {"tflops": tf, "ms": ms, "effective_clock_ghz": tf * 1e12 / (1024 * 1024 * 1e9)}


def cpu_baseline(batch=8, size=224, threads=16, timed_steps=3):
    """The reference's CPU algorithm (oracle/, pinned to the reference by golden vectors) on the host cores:
    one warm-up + one timed SP train step (fwd + floss + bwd + Adam) at a bounded batch."""
    from oracle import egaze_oracle as O
    from oracle import synth
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # torch-CPU convolution training does not scale past ~16 threads on the 2x64-core EPYC host of the MI355X
    # box (measured, tools/cpu_baseline_sweep.py: 16 thr 3.2 frames/s, 32 thr 2.8, 64 thr 1.4, 256 thr > 5 min/step),
    # so the baseline uses the best-performing thread count and reports it as `cores`.
    cores = min(threads, avail)
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: {cores} of {avail} host threads, batch {batch}", file=sys.stderr, flush=True)
    sd = synth.synth_state_dict(O.sp_shapes(), seed=1, head_gain=0.25)
    x_s, x_t, gt, _ = synth.synth_sp_batch(batch, size, seed=0)
    opt = {}
    O.sp_train_step(sd, opt, 1, x_s, x_t, gt, 1e-7)
    t0 = time.perf_counter()
    for i in range(timed_steps):
        O.sp_train_step(sd, opt, 2 + i, x_s, x_t, gt, 1e-7)
    dt = (time.perf_counter() - t0) / timed_steps
    return {"value": batch / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle SP train step (fwd+floss+bwd+Adam; the reference's PyTorch-CPU algorithm), batch "
                      f"{batch}, {size}x{size}, 1 warm-up + {timed_steps} timed steps, {dt:.2f} s/step, torch-CPU "
                      f"fp32 on {cores} threads (best of an 8/16/32/64 sweep, profiles/r06_cpu_baseline_thread_sweep.txt; host has "
                      f"{avail}); the same oracle step at batch 32 was measured once per round on a GPU box "
                      f"(profiles/r06_cpu_baseline_b32.txt) -- batch 8 is the bounded sample, and the faster of the two per frame"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region (--steps steps between barrier + synchronize) is run this many times back to back; "
                         "value / ms_per_step are the MEDIAN region, extra.timed_repeats lists all of them")
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU (BASELINE: 32)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--no-at", action="store_true", help="leave the AT (lstmnet T=16, B=32) training step out")
    ap.add_argument("--at-form", choices=("wave", "persist"), default="wave",
                    help="form of the AT recurrence INSIDE the combined step: the wavefront launches (default: they share the chip "
                         "with the SP kernels) or the persistent weight-stationary launches (A/B runs; stand-alone AT always times the latter)")
    ap.add_argument("--no-input-prefetch", action="store_true",
                    help="A/B: convert the flow stack to its NHWC-32 form at the head of the forward pass instead of one step ahead on a helper stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the untimed exact-f32-mode step timing")
    ap.add_argument("--no-bwd-leg", action="store_true",
                    help="skip the opt-in two-product backward leg (extra.bwd2): every step of the run is then the default arithmetic "
                         "(what the committed kernel statistics / PMC profiles are collected with)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU.
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] --gpus {args.gpus}: launching {args.gpus} ranks under torch.distributed.run", file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    backend = os.environ.get("EGAZE_DIST_BACKEND", "nccl")       # "gloo": functional DP run, e.g. N ranks on a 1-GPU box
    ndev = torch.cuda.device_count()
    shared_device = os.environ.get("EGAZE_SINGLE_DEVICE") == "1"
    if shared_device:
        local = 0                 # all ranks on GPU 0: a FUNCTIONAL data-parallel run (needs EGAZE_DIST_BACKEND=gloo)
        if backend == "nccl":
            raise SystemExit("EGAZE_SINGLE_DEVICE=1 needs EGAZE_DIST_BACKEND=gloo (RCCL wants one device per rank)")
    elif world > ndev:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} visible GPU(s); set EGAZE_SINGLE_DEVICE=1 "
                         f"EGAZE_DIST_BACKEND=gloo for a functional run of the data-parallel path on one GPU")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # EGAZE_DP_FORCE=1 at N = 1: a process group of ONE rank over RCCL -- the timed step then runs the data-parallel code
    # path (bucket hooks, async all-reduce from the comm stream, the joins in front of Adam) on the one GPU of the box
    dp_forced = world == 1 and os.environ.get("EGAZE_DP_FORCE") == "1"
    if world > 1 or dp_forced:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dp_forced and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()               # the first collective creates the communicator (and prints RCCL's banner)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
        if rank == 0:
            print(f"[bench] {'RCCL' if backend == 'nccl' else backend} ranks: {dist.get_world_size()}"
                  f"{' (all on one GPU: functional run)' if shared_device else ''}", file=sys.stderr, flush=True)

    import egaze_amd  # noqa: F401
    import egaze_amd.hipops as H
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.utils import make_layers, cfg
    from egaze_amd.floss import floss
    from egaze_amd.optim import FusedAdam
    from egaze_amd import dp, streams, synthetic

    torch.manual_seed(1234)
    model = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev)   # random init (no weights offline)
    model.train()
    criterion = floss().to(dev)
    optimizer = FusedAdam(model.parameters(), lr=1e-7)          # gaze_full.py --lr default
    if dist is not None:
        dp.attach(optimizer)
    batch = synthetic.sp_batch(args.batch, args.size, dev, seed=100 + rank)
    input_s, input_t, target = batch["image"], batch["flow"], batch["gt"]

    # AT module (BASELINE config 4 shape): lstmnet over T=16 steps of 512-vectors, batch 32, explicit (h, c)
    use_at = not args.no_at
    if use_at:
        from egaze_amd.models.LSTMnet import lstmnet
        from egaze_amd.functions import MSELoss
        T_AT = 16
        lstm = lstmnet().to(dev)
        lstm.train()
        opt_at = FusedAdam(lstm.parameters(), lr=1e-4)          # AT.py:84
        if dist is not None:
            dp.attach(opt_at)
        atb = synthetic.at_batch(T_AT, args.batch, dev, seed=200 + rank)
        at_in, at_tgt = atb["input"], torch.tanh(atb["gt"])
        h0 = torch.zeros(2, args.batch, 512, device=dev)
        c0 = torch.zeros(2, args.batch, 512, device=dev)
        opt_at.zero_grad()

    def at_step():
        # AT: forward + MSE + backward + Adam of the attention-transition LSTM (AT.py:138-145 at T=16, B=32)
        pred, _ = lstm(at_in, (h0, c0))
        l2 = MSELoss.apply(pred, at_tgt)
        l2.backward()
        opt_at.step()
        opt_at.zero_grad()

    # The AT module is a separate model on separate inputs (the reference trains it from extracted features, AT.py):
    # its ~200 small, launch-latency-bound kernels are issued on their own HIP stream and run in the shadow of the SP
    # step's large kernels.  Stream order keeps AT step k+1 behind AT step k; the timed region ends with a device-wide
    # synchronize, so every AT step is complete inside it.  EGAZE_STREAMS=0 serialises it again.
    at_stream = None
    if use_at and streams.ENABLED:
        at_stream = streams.side_stream("at")
        at_stream.wait_stream(torch.cuda.current_stream())

    prefetch_stream = streams.side_stream("input") if (streams.ENABLED and not args.no_input_prefetch) else None

    def step(staged=None):
        # SP: the body of SP.trainSP's loop (SP.py:132-138); ``staged`` = (image, flow, gt) of this step when the batch
        # crosses PCIe inside the step (the untimed PCIe-inclusive leg below), else the HBM-resident synthetic batch
        x_s, x_t, tgt = staged if staged is not None else (input_s, input_t, target)
        output = model(x_s, x_t)
        if prefetch_stream is not None and staged is None and streams.ENABLED:
            # the NEXT step's input-side work (flow stack -> NHWC-32 + its abs-max: no weights involved), issued now on a helper
            # stream exactly as SP.trainSP's loader does for batch k + 1 (data.STdatas.staged_batches): it runs under this step's
            # backward pass instead of at the serial head of the next forward pass.  Still once per step, inside the timed region.
            with torch.cuda.stream(prefetch_stream):
                H.prepare_network_input(x_t)
        loss = criterion(output, tgt.view(output.size()))
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
        if use_at:
            if at_stream is not None and streams.ENABLED:
                # beside the SP step's kernels: the wavefront form (the persistent LSTM kernels want every CU to themselves; the
                # stand-alone AT leg below -- BASELINE config 4 -- runs them)
                with torch.cuda.stream(at_stream), H.lstm_persistent(args.at_form == "persist"):
                    at_step()
            else:
                at_step()
        return loss

    probe = None
    if rank == 0 and not args.no_roofline:
        probe = mfma_power_probe(dev)
        print(f"[bench] bare-MFMA probe (f16 32x32x16, random bits): {probe['tflops']:.0f} TFLOP/s "
              f"= {probe['effective_clock_ghz']:.2f} GHz effective", file=sys.stderr, flush=True)
    optimizer.zero_grad()
    for _ in range(args.warmup):
        step()
    # The timed region: exactly --steps steps between barrier + synchronize on both sides, MAX over ranks.  It is run
    # --repeats times back to back in this process (a 20-step region is 0.6 s and one region per round cannot resolve the
    # sub-1 % changes the round's A/B notes argue about: VERDICT r4): value = the MEDIAN region, all regions reported.
    def timed_regions():
        """--repeats regions of exactly --steps steps, each bracketed by barrier + synchronize on both sides, MAX over ranks."""
        regs, last = [], None
        for _ in range(max(1, args.repeats)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                last = step()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = t.item()
            regs.append(el)
        return regs, last

    def median(regs):
        return sorted(regs)[(len(regs) - 1) // 2]

    regions, loss = timed_regions()
    elapsed = median(regions)
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = args.batch * world * args.steps / elapsed
    last_loss = loss.item()
    del loss            # the last graph (and its AccumulateGrad nodes, bound to this leg's streams) dies here

    roofline = None
    breakdown = None
    f32_ms = None
    other = None
    at_ms = None
    at_eager_ms = None
    at_graph_error = None
    pcie_ms = {}
    rccl = None
    lf_block = None
    if not args.no_roofline:
        # Every rank runs these two extra (untimed) steps -- the gradient all-reduce inside step() is a collective --
        # but only rank 0 reports.  Per-kernel HIP-event timing needs the kernels serialised: the multi-stream
        # overlap is switched off so that an event pair brackets exactly one kernel family's launches.
        def profiled_step():
            """One step with per-kernel-family HIP-event timing (hipops.PROF).  The kernels must be serialised for an event pair
            to bracket exactly one family's launches: the multi-stream overlap is switched off for this step."""
            was = streams.ENABLED
            streams.ENABLED = False
            step()
            torch.cuda.synchronize()
            H.PROF.start()
            step()
            pr = H.PROF.stop()
            streams.ENABLED = was
            return pr

        split = H.PRECISION == "split"

        def conv_family_roofline(pr):
            ig = {"calls": 0, "ms": 0.0, "flops": 0.0}
            # split mode: every conv fwd / dgrad launch runs on the streamed-weight kernel (egz_conv3x3_fwd_streamed; the per-tap
            # gather kernel egz_conv3x3_fwd_split only for geometries it does not cover); hipops notes the algorithmic FLOPs of the
            # whole family under ONE name (egz_conv3x3_fwd_split), whether or not that entry point itself ran in the step
            entries = ("egz_conv3x3_fwd_split", "egz_conv3x3_fwd_streamed", "egz_conv3x3_fwd_streamed_splitk") if split else ("egz_conv3x3_fwd", "egz_conv3x3_ups_dgrad")
            for entry in entries:
                for k2 in ig:
                    ig[k2] += pr.get(entry, {}).get(k2, 0)
            if ig["ms"] <= 0:
                return None
            p2_flops = pr.get("two_product_conv", {}).get("flops", 0.0)
            mfma_per_mac = 3.0 - min(1.0, p2_flops / ig["flops"]) if ig["flops"] else 3.0
            # HBM bytes per launch of this kernel family: PMC counters cannot be sampled from inside the process, so the value
            # comes from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/collect_profiles.sh
            # -> tools/pmc_traffic.py; gfx950 FETCH half-count corrected), collected with --no-bwd-leg --no-f32-leg so that every
            # profiled step is the default arithmetic.  The profile is stamped with a hash of the kernel sources it was taken
            # from: if the kernels changed since, the number is NOT quoted (traffic = null).
            traffic, traffic_note = None, None
            tpath = os.path.join(ROOT, "profiles", "r06_pmc_traffic_conv_fwd_dgrad.json" if split else "r01_pmc_traffic_igemm.json")
            if p2_flops:
                traffic_note = "the committed PMC profile is of the default (three-product) arithmetic"
            elif os.path.exists(tpath):
                tj = json.load(open(tpath))
                if not split or tj.get("kernel_src_sha") == kernel_src_sha():
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_note = f"profiles/{os.path.basename(tpath)}"
                else:
                    traffic_note = (f"profiles/{os.path.basename(tpath)} was taken from other kernel sources "
                                    f"({tj.get('kernel_src_sha')} vs {kernel_src_sha()}): re-run tools/collect_profiles.sh")
            achieved = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
            peak = F16_MFMA_PEAK_TFLOPS if split else F32_MFMA_PEAK_TFLOPS
            rf = {"bound": "mfma",
                  "kernel": ("conv3x3_igemm_x3s_kernel (egz_conv3x3_fwd_streamed, + egz_conv3x3_fwd_split for "
                             "geometries it does not cover: all conv fwd + dgrad launches -- the streamed-weight halo "
                             "kernel for plain convs, the four-phase upsample forward and the polyphase upsample "
                             "dgrad; split-half f16x3 / bf16x3 operands on "
                             "v_mfma_f32_32x32x16_{f16,bf16}; an algorithmic MAC costs mfma_macs_per_algorithmic_mac MFMA MACs "
                             "(3 = three products in forward and data-gradient launches alike, hipops.BWD_PRODUCTS), priced "
                             "against the dense 16-bit MFMA peak)" if split else
                             "conv3x3_igemm_kernel (egz_conv3x3_fwd + egz_conv3x3_ups_dgrad: all fwd + dgrad "
                             "launches, exact-f32 MFMA)") +
                            "; FLOPs are the reference's algorithmic count, the upsample-fused launches execute 4/9 of it; "
                            "timed with HIP events on the launch stream with stream concurrency off, as in "
                            "profiles/r06_bench_b32_kernel_stats_streams0.txt",
                  "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                  "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_note,
                  "achieved_vs_f32_mfma_peak": achieved / F32_MFMA_PEAK_TFLOPS,
                  "launches_per_step": ig["calls"], "avg_launch_ms": ig["ms"] / ig["calls"],
                  "algorithmic_flop_per_step": ig["flops"]}
            if split:
                rf["mfma_macs_per_algorithmic_mac"] = mfma_per_mac
            if probe is not None and split:
                # this box's own ceiling: bare MFMA rate on random bits / MFMA MACs per algorithmic MAC of this family
                rf.update({"power_ceiling_tflops": probe["tflops"], "effective_clock_ghz": probe["effective_clock_ghz"],
                           "frac_of_power_ceiling": achieved / (probe["tflops"] / mfma_per_mac),
                           "power_ceiling_note": "egz_mfma_probe (csrc/probe.hip) run for ~30 ms before the timed region: "
                                                 "sustained v_mfma_f32_32x32x16_f16 TFLOP/s on random operand bits; "
                                                 "frac_of_power_ceiling = achieved / (ceiling / mfma_macs_per_algorithmic_mac)"})
            return rf

        def ms_breakdown(pr):
            bd = {k: round(v["ms"], 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"]) if v["calls"]}
            bd["_sum_kernel_ms"] = round(sum(v["ms"] for v in pr.values()), 3)
            return bd

        # Every rank runs these extra (untimed) steps -- the gradient all-reduce inside step() is a collective -- but only
        # rank 0 reports.
        prof = profiled_step()
        roofline = conv_family_roofline(prof)
        breakdown = ms_breakdown(prof)
        if use_at:
            # config 4 standalone: the AT step alone (lstmnet T=16, B=32 forward + MSE + backward + Adam), untimed leg
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                at_step()
            torch.cuda.synchronize()
            at_eager_ms = (time.perf_counter() - t1) / 20 * 1e3
            at_ms = at_eager_ms
            if dist is None:
                # the same step captured into ONE hipGraph and replayed (graphs.GraphedTrainStep -- what LF._run does for its
                # iteration and AT.trainLSTM for its per-sample step): with the recurrence in two persistent launches the eager
                # step is bound by the host issuing its ~40 launches and autograd nodes, not by the device
                from egaze_amd.graphs import GraphedTrainStep

                def at_forward_loss(x, tgt):
                    pred, _ = lstm(x, (h0, c0))
                    return MSELoss.apply(pred, tgt), pred
                gat = GraphedTrainStep(at_forward_loss, opt_at, (at_in, at_tgt))
                try:                       # (a side leg must not lose the headline line: on failure the eager figure stands)
                    for _ in range(6):
                        gat(at_in, at_tgt)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(50):
                        gat(at_in, at_tgt)
                    torch.cuda.synchronize()
                    at_ms = (time.perf_counter() - t1) / 50 * 1e3
                except Exception as e:
                    at_graph_error = repr(e)[:300]
                    print(f"[bench] graphed AT leg failed: {at_graph_error}", file=sys.stderr, flush=True)
                finally:
                    gat.close()
            # the persistent recurrence launches of this leg report lost hand-offs through sticky words; a bench that timed a broken
            # step must say so
            try:
                H.lstm_persist_check()
                opt_at.check_finite()
            except Exception as e:
                at_graph_error = ((at_graph_error + "; ") if at_graph_error else "") + repr(e)[:300]
                at_ms = at_eager_ms = None          # (not a number to quote)
                print(f"[bench] AT leg: {at_graph_error}", file=sys.stderr, flush=True)
        if world == 1:
            # BASELINE config 5's last stage beside the headline (untimed leg): one LF.trainLate iteration (late_fusion forward
            # + floss + backward + Adam, LF.py:90-100) at the same batch, HBM-bound -- 88 MB of algorithmic traffic per frame
            # (every conv reads its input and writes its output once, BN / ReLU fused) against the 8 TB/s roof
            try:
                from egaze_amd.models.late_fusion import late_fusion
                lfm = late_fusion().to(dev)
                lfm.train()
                lfo = FusedAdam(lfm.parameters(), lr=1e-4)
                lfb = [torch.rand(args.batch, 1, args.size, args.size, device=dev) for _ in range(3)]

                def lf_step():
                    o = lfm(lfb[0], lfb[1])
                    l_ = criterion(o, lfb[2])
                    lfo.zero_grad()
                    l_.backward()
                    lfo.step()
                for _ in range(3):
                    lf_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    lf_step()
                torch.cuda.synchronize()
                lf_eager_ms = (time.perf_counter() - t1) / 20 * 1e3
                # ... the same model step captured into one hipGraph (graphs.GraphedTrainStep) ...
                from egaze_amd.graphs import GraphedTrainStep
                lfg = GraphedTrainStep(lambda a_, b_, c_: (criterion(lfm(a_, b_), c_),), lfo, tuple(lfb))
                for _ in range(3):
                    lfg(*lfg.static_in)             # (its own input buffers: a resident batch, no staging copy)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(40):
                    lfg(*lfg.static_in)
                torch.cuda.synchronize()
                lf_ms = (time.perf_counter() - t1) / 40 * 1e3
                lfg.close()
                # ... and the FULL LF.trainLate iteration as LF._run issues it at world size 1 (LF.GraphedLateIteration): the
                # reference's loop also evaluates computeAAEAUC on every batch and feeds loss / AAE / AUC into running averages
                # (LF.py:90-100) -- the metric kernel sits inside the captured step, the values are read back 16 iterations at a time
                from egaze_amd.LF import GraphedLateIteration
                lfi = GraphedLateIteration(lfm, criterion, lfo, (lfb[0], lfb[1], lfb[2]))
                got = []
                for _ in range(4):
                    lfi(*lfi.step.static_in)
                got += lfi.drain()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(48):
                    lfi(*lfi.step.static_in)
                    if lfi.full:
                        got += lfi.drain()
                got += lfi.drain()
                torch.cuda.synchronize()
                lf_full_ms = (time.perf_counter() - t1) / 48 * 1e3
                lfi.close()
                assert len(got) == 52
                lf_bytes = 88e6 * args.batch * (args.size / 224.0) ** 2

                def _roof(ms):
                    return {"bound": "hbm", "achieved": lf_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": lf_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": 88e6}
                lf_block = {"iteration_ms": lf_full_ms, "iteration_frames_per_s": args.batch / (lf_full_ms * 1e-3),
                            "roofline": _roof(lf_full_ms),
                            "model_step_ms": lf_ms, "model_step_frames_per_s": args.batch / (lf_ms * 1e-3),
                            "model_step_roofline_frac": _roof(lf_ms)["frac"],
                            "eager_model_step_ms": lf_eager_ms,
                            "last_iteration": {"loss": got[-1][0], "aae_deg": got[-1][1], "auc": got[-1][2]},
                            "note": "untimed leg at this batch.  iteration = LF.trainLate's whole loop body (LF.py:90-100: "
                                    "late_fusion forward, floss, computeAAEAUC of the batch, zero_grad, backward, Adam) as ONE "
                                    "hipGraph replay, loss / AAE / AUC read back 16 iterations at a time, 48 iterations timed; "
                                    "`roofline` prices THIS leg (88 MB of algorithmic traffic per frame, SURVEY.md 8d).  "
                                    "model_step = the same without the metric (40 replays), eager_model_step = launch by launch "
                                    "(20 steps)"}
                del lfm, lfo, lfb, lfg, lfi
            except Exception as e:
                lf_block = {"error": repr(e)[:300]}
        if split and not args.no_bwd_leg and H.GRAD_SPLIT == "f16":
            # The OTHER backward arithmetic with the SAME protocol as the headline (--warmup steps, --repeats regions of --steps
            # steps between barrier + synchronize, median): by default the headline is three products per MAC (fp32-class
            # gradients) and this leg the opt-in two-product backward (EGAZE_BWD_PRODUCTS=2), reported as extra.bwd2 with its
            # own roofline -- never as `value`.
            headline_products = H.BWD_PRODUCTS
            H.BWD_PRODUCTS = 2 if headline_products == 3 else 3
            for _ in range(max(args.warmup, 2)):
                step()
            o_regions, _ = timed_regions()
            o_prof = profiled_step()
            other = {"bwd_products": H.BWD_PRODUCTS,
                     "ms_per_step": median(o_regions) / args.steps * 1e3,
                     "value": args.batch * world * args.steps / median(o_regions), "unit": "frames/s",
                     "regions_ms_per_step": [r / args.steps * 1e3 for r in o_regions],
                     "roofline": conv_family_roofline(o_prof),
                     "kernel_ms_breakdown": ms_breakdown(o_prof),
                     "note": ("OPT-IN arithmetic, not the headline: EGAZE_BWD_PRODUCTS=2 -- the backward convolutions issue two MFMA "
                              "products per MAC, one operand of every backward product enters with 11 significant bits "
                              "(per-op gradient error 2e-4 relative L2, <= 1e-3 through the chain; forward / gaze map "
                              "bit-identical); same timing protocol as the headline" if H.BWD_PRODUCTS == 2 else
                              "the default arithmetic (three products per MAC), same timing protocol as the headline, which "
                              "was run with EGAZE_BWD_PRODUCTS=2 set")}
            H.BWD_PRODUCTS = headline_products
            for _ in range(2):
                step()
        if split and not args.no_f32_leg:
            # the same step on the exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32), untimed leg, reported beside the headline
            H.PRECISION = "f32"
            for _ in range(2):
                step()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            f32_ms = (time.perf_counter() - t1) / 3 * 1e3
            H.PRECISION = "split"
        if world == 1:
            # PCIe-inclusive step: the batch starts in pinned host memory and is staged inside the step the way SP.trainSP
            # does it (data.STdatas.stage_batch): raw bytes + device-side normalisation (this build's loader, 38.5 MB per
            # batch) and the reference loader's normalised fp32 tensors (154 MB per batch).  Untimed leg, never `value`.
            from egaze_amd.data.STdatas import stage_batch, staged_batches
            g = torch.Generator().manual_seed(7)
            shapes = {"image": (args.batch, 3, args.size, args.size), "flow": (args.batch, 20, args.size, args.size),
                      "gt": (args.batch, 1, args.size, args.size)}
            for tag in ("u8", "f32"):
                if tag == "u8":
                    host = {k: torch.randint(0, 256, sh, dtype=torch.uint8, generator=g).pin_memory() for k, sh in shapes.items()}
                else:
                    host = {k: torch.rand(sh, generator=g).pin_memory() for k, sh in shapes.items()}
                # (a) copied inside the step on the compute stream, as the reference's loop does (SP.py:126-131)
                for _ in range(2):
                    step(stage_batch(host, dev))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    step(stage_batch(host, dev)).item()            # the driver reads the loss every step (SP.py:139)
                pcie_ms[tag + "_in_step"] = (time.perf_counter() - t1) / 5 * 1e3
                # (b) this build's SP.trainSP: batch k + 1 staged on a copy stream while step k computes
                t1 = None
                for k, (_, staged) in enumerate(staged_batches([host] * 7, dev)):
                    if k == 2:
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                    step(staged).item()
                pcie_ms[tag + "_prefetched"] = (time.perf_counter() - t1) / 5 * 1e3
        if world == 1 and dist is None:
            # The data-parallel code path on this one GPU (untimed leg): a process group of ONE rank over RCCL, the gradient
            # reducer attached to the live optimizers -- bucket hooks fired from the gradient sinks during backward, async
            # all-reduces issued from the comm stream, handles joined in front of Adam -- against the same steps without it.
            redirect = stdout_to_stderr()
            redirect.__enter__()
            try:
                import socket
                import torch.distributed as tdist
                def _timed(n, reps=3):
                    # the fastest of `reps` windows of n steps: a 5-step window scatters by +-0.3 ms on these boxes (the first
                    # window of a leg also carries its one-off allocations), as large as the difference this leg is after
                    best = None
                    for _ in range(reps):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(n):
                            step()
                        torch.cuda.synchronize()
                        dt = (time.perf_counter() - t1) / n * 1e3
                        best = dt if best is None or dt < best else best
                    return best
                step()
                plain_ms = _timed(6)
                with socket.socket() as s_:
                    s_.bind(("127.0.0.1", 0))
                    port = s_.getsockname()[1]
                tdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
                tdist.barrier()                  # creates the communicator
                # the two legs that differ by the reducer only, ALTERNATING (three rounds, the fastest window of each): legs run one
                # after the other drift apart by a few tenths of a millisecond with the chip's power state, as much as the difference
                group_only_ms = rccl_ms = None
                for _ in range(3):
                    step()
                    t_ = _timed(6, reps=1)           # an RCCL communicator exists, the reducer does not: RCCL's own side effect
                    group_only_ms = t_ if group_only_ms is None or t_ < group_only_ms else group_only_ms
                    reds = [(dp.attach(optimizer, force=True), optimizer)]
                    if use_at:
                        reds.append((dp.attach(opt_at, force=True), opt_at))
                    step()
                    step()
                    t_ = _timed(6, reps=1)
                    rccl_ms = t_ if rccl_ms is None or t_ < rccl_ms else rccl_ms
                    nb = len(reds[0][0].buckets)
                    inb = reds[0][0].stats["launched_in_backward"] / max(reds[0][0].stats["steps"], 1)
                    for red, o in reds:
                        red.detach(o)
                tdist.destroy_process_group()
                rccl = {"ms_per_step": rccl_ms, "ms_per_step_without": plain_ms, "ms_per_step_group_initialised_reducer_off": group_only_ms,
                        "delta_ms": rccl_ms - plain_ms, "delta_ms_of_the_reducer": rccl_ms - group_only_ms,
                        "buckets": nb, "buckets_issued_inside_backward_per_step": inb,
                        "note": "untimed leg, three legs (the fastest of three 6-step windows each; the group-only and reducer legs alternate) on the live optimizers: no process group / an RCCL group "
                                "of one rank initialised and nothing attached / dp.GradReducer forced on (bucket hooks fired from "
                                "the gradient sinks, all-reduces issued from the comm stream inside backward, joined in front of "
                                "Adam).  delta_ms = reducer on - no group, delta_ms_of_the_reducer = reducer on - group only.  "
                                "The timed region above runs without any of it at N = 1"}
            except Exception as e:       # a box without a working RCCL must not lose the bench line
                rccl = {"error": repr(e)[:300]}
            finally:
                redirect.__exit__()
    if dist is not None:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    weak = None
    if rank == 0:
        # Weak-scaling bookkeeping across the driver's back-to-back runs (N = 1, 2, 4, 8 on one node): the N = 1 run leaves its
        # line's key numbers in $TMPDIR, an N > 1 run of the same per-GPU workload on the same host reports its step time against
        # it.  (The driver computes efficiency ITSELF from the per-N values; this block is a convenience for whoever reads one line.)
        import socket as _so
        import tempfile
        n1_path = os.path.join(os.environ.get("TMPDIR") or tempfile.gettempdir(), "egaze_bench_n1.json")
        sig = {"batch": args.batch, "size": args.size, "at": use_at, "host": _so.gethostname(), "bwd_products": H.BWD_PRODUCTS,
               "precision": H.PRECISION, "steps": args.steps}
        if world == 1 and not dp_forced:
            try:
                json.dump({"sig": sig, "ms_per_step": ms_per_step, "frames_per_s": frames_per_s}, open(n1_path, "w"))
            except OSError:
                pass
        elif world > 1 and os.path.exists(n1_path):
            try:
                n1 = json.load(open(n1_path))
                if n1.get("sig") == sig:
                    weak = {"n1_ms": n1["ms_per_step"], "this_ms": ms_per_step, "efficiency": n1["ms_per_step"] / ms_per_step,
                            "n1_frames_per_s": n1["frames_per_s"], "speedup": frames_per_s / n1["frames_per_s"],
                            "note": f"per-GPU workload fixed (weak scaling): efficiency = N=1 step time / this run's step time; "
                                    f"the N = 1 line was saved by an earlier run of this script on this host ({n1_path})"}
            except (OSError, ValueError, KeyError):
                weak = None
    if rank == 0:
        step_flops = FLOP_PER_FRAME_FWD_BWD * args.batch * (args.size / 224.0) ** 2
        out = {
            "metric": "SP+AT train frames/sec/node (224x224, bs32/GPU)",
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (every convolution -- forward, data gradient, weight gradient -- as "
                      + ("f16 split-half MFMA, abs-max scaled operands, " + ("THREE products per MAC: 22 significant bits on both operands of every product"
                         if H.BWD_PRODUCTS == 3 else "three products per MAC forward; OPT-IN EGAZE_BWD_PRODUCTS=2 backward: 22 x 11 significant bits")
                         if H.GRAD_SPLIT == "f16" else "f16x3 forward, bf16x3 (16 bits) backward: OPT-IN EGAZE_GRAD_SPLIT=bf16")
                      + "; fp32 accumulate; everything else exact f32)" if H.PRECISION == "split" else "f32"),
            "data": "synthetic",
            "config": {"workload": f"SP two-stream (RGB + 10-pair flow stack) forward + floss + backward + Adam, "
                                   f"batch {args.batch}/GPU, {args.size}x{args.size}, train-mode BN, all "
                                   f"46.5M params trainable (--sp_resume 0)"
                                   + (f"; + AT lstmnet forward + MSE + backward + Adam over T=16, B={args.batch} "
                                      f"512-vectors per step, issued on a side stream beside the SP kernels in its WAVEFRONT "
                                      f"form (one launch per (layer, step) diagonal; the persistent weight-stationary launches "
                                      f"that extra.at_ms_per_step times stand-alone want every CU to themselves)" if use_at else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "collective": ("gradient all-reduce over rccl at world size 1 (EGAZE_DP_FORCE=1: the data-parallel code "
                                      "path on one GPU)" if dp_forced else None if world == 1 else
                                      f"gradient all-reduce, backend {'rccl' if backend == 'nccl' else backend}, "
                                      f"{'ALL RANKS ON ONE GPU (functional run, not a scaling number)' if shared_device else 'one GPU per rank'}"),
                       "precision": (f"split-half f16x3 MFMA for conv fwd, {'f16 halves' if H.GRAD_SPLIT == 'f16' else 'bf16x3'} "
                                     f"for dgrad / wgrad ({'three products per MAC (fp32-class gradients, 2e-7 per op; extra.bwd2 = the opt-in two-product backward, same protocol)' if H.BWD_PRODUCTS == 3 else 'OPT-IN two products per MAC: one operand of every backward product enters with 11 significant bits; extra.bwd3 = the default'}), "
                                     "operands abs-max scaled (forward fp32-class: gaze map within 1e-5 of "
                                     "the reference at batch 2 and batch 32; an 8-step lr 1e-4 training trajectory: per-step loss inside "
                                     "4x, end state (eval gaze map, BN statistics) inside 2x the CPU fp32 path's own distance from an "
                                     "fp64 run of the same steps -- such a trajectory is chaotic at the 1e-3 level for any fp32 "
                                     "implementation, profiles/r03_training_trajectory.txt; "
                                     "tests/test_hip_model_sp.py)"
                                     if H.PRECISION == "split" else "exact f32 MFMA (v_mfma_f32_32x32x2_f32)")},
            "roofline": roofline, "cpu_baseline": cpu,
            # algorithmic step FLOPs / s against the exact-f32 MFMA peak (157.3 TF/s): NOT a utilisation -- the default mode runs
            # 16-bit MFMAs (3 per algorithmic MAC), so the ratio exceeds 1; the f32 mode's own figure is f32_ms_per_step
            "step_algorithmic_tflops": step_flops / (ms_per_step * 1e-3) / 1e12,
            "step_algorithmic_tflops_over_f32_mfma_peak": step_flops / (ms_per_step * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
            "step_hbm_frac": BYTES_PER_FRAME * args.batch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "loss": last_loss, "kernel_ms_breakdown": breakdown,
            "extra": {"timed_repeats": {"regions": len(regions), "steps_per_region": args.steps,
                                        "ms_per_step": [r / args.steps * 1e3 for r in regions],
                                        "min_ms_per_step": min(regions) / args.steps * 1e3,
                                        "max_ms_per_step": max(regions) / args.steps * 1e3,
                                        "value_is": "the median region (value, ms_per_step); every region is --steps steps "
                                                    "bracketed by barrier + synchronize, max over ranks"},
                      "at_ms_per_step": at_ms,
                      "at_eager_ms_per_step": at_eager_ms,
                      "at_graph_error": at_graph_error,
                      "at_roofline": (None if not at_ms else
                                      {"bound": "latency (5 batched f32-MFMA GEMM launches + the recurrence as two persistent "
                                                "weight-stationary launches: 17 forward / 18 backward in-launch steps of ~5 / ~6.6 us, "
                                                "each an all-to-all of h / dgates between 256 resident blocks)",
                                       "achieved": 3 * 4.56e9 * (args.batch / 32.0) / (at_ms * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": 3 * 4.56e9 * (args.batch / 32.0) / (at_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                       "samples_per_s": 16 * args.batch / (at_ms * 1e-3),
                                       "note": "config 4 shape, T=16: 4.56 GFLOP forward, x3 for forward + backward (SURVEY.md 8d); "
                                               "one hipGraph replay per step (at_eager_ms_per_step: the same step issued "
                                               "launch by launch, host-bound); not a matrix-core workload"}),
                      "lf_step": lf_block,
                      "at_note": ("AT alone (BASELINE config 4 shape): lstmnet T=16, B=%d forward + MSE + backward + Adam, "
                                  "%s (t, b) samples/s" % (args.batch, ("%.0f" % (16 * args.batch / (at_ms * 1e-3))) if at_ms else "n/a")),
                      ("bwd2" if other is None or other["bwd_products"] == 2 else "bwd3"): other,
                      "f32_ms_per_step": f32_ms,
                      "f32_note": "same step with EGAZE_PRECISION=f32 (exact-f32 MFMA everywhere), 3 untimed-leg steps",
                      "rccl_world1": rccl,
                      "weak_scaling": weak,
                      "pcie_inclusive_ms_per_step": pcie_ms or None,
                      "pcie_note": ("the step with its batch starting in pinned host memory and the loss read back every step "
                                    "(SP.trainSP's loop): 'u8' = raw bytes + egz_u8_normalize on the device (38.5 MB/batch), "
                                    "'f32' = the reference loader's normalised fp32 tensors (154 MB/batch); '_in_step' = "
                                    "copied on the compute stream inside the step (the reference's loop), '_prefetched' = "
                                    "batch k+1 staged on a copy stream during step k (data.STdatas.staged_batches, what "
                                    "SP.trainSP does); 5 untimed-leg steps each")},
        }
        print(json.dumps(out))
    if dist is not None and os.environ.get("EGAZE_DP_CHECK") == "1":
        # functional check of the data-parallel path: replicas must hold identical parameters after the steps
        ref = optimizer.flat_p.clone()
        dist.broadcast(ref, src=0)
        same = torch.equal(ref, optimizer.flat_p)
        print(f"[dp-check] rank {rank}: parameters identical to rank 0: {same}; loss {last_loss:.6f}", flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
