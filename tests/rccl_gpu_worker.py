"""Worker of tests/test_hip_rccl.py: the `nccl` (= RCCL) backend at world size 1 on the one GPU of the test box.

A 1-rank all-reduce is legal and runs exactly the code path the 8-GPU job runs per rank: dp.GradReducer's hooks fire from
the gradient sinks during backward, every bucket is issued as an ASYNC RCCL all-reduce from the comm stream (which waits
for all producer streams), and FusedAdam.step() joins the handles before the Adam kernel.  gloo -- the backend of every
other DP test -- stages device tensors through the host and synchronises, so none of that ordering is exercised there.
Writes its observations as JSON to argv[1]."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, size, batch, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    import egaze_amd  # noqa: F401
    import egaze_amd.hipops as H
    from egaze_amd import dp, streams, synthetic
    from egaze_amd.floss import floss
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.optim import FusedAdam
    from egaze_amd.utils import cfg, make_layers

    assert H.PRECISION == "split" and streams.ENABLED, "the test is about the default (split-half, streams on) path"
    torch.manual_seed(1234)
    model = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev)
    model.train()
    crit = floss().to(dev)
    opt = FusedAdam(model.parameters(), lr=1e-4)
    b = synthetic.sp_batch(batch, size, dev, seed=100)
    p0 = opt.flat_p.clone()

    def restore():
        opt.flat_p.copy_(p0)
        opt.flat_m.zero_()
        opt.flat_v.zero_()
        opt.step_count = 0
        H.bump_weight_epoch()

    def fwd_bwd():
        opt.zero_grad()
        out = model(b["image"], b["flow"])
        loss = crit(out, b["gt"].view(out.size()))
        loss.backward()
        return loss

    # (0) no reducer: the gradient of one backward pass and the parameters after two optimizer steps
    fwd_bwd()
    torch.cuda.synchronize()      # (backward() itself now joins the helper streams: functions._join_at_end_of_backward)
    g_ref = opt.flat_g.clone()
    for _ in range(2):
        fwd_bwd()
        opt.step()
    torch.cuda.synchronize()
    p_ref = opt.flat_p.clone()

    # (1) RCCL reducer forced on at world 1, 8 MB buckets, HIP events behind every bucket's collective
    restore()
    red = dp.attach(opt, bucket_bytes=8 * 1024 * 1024, force=True, record_events=True)
    assert red.active and red.world == 1 and dist.get_backend() == "nccl"
    fwd_bwd()
    ev_bwd_end = torch.cuda.Event(enable_timing=True)
    ev_bwd_end.record()                                    # the stream position right behind the last backward kernel
    in_backward = red.stats["launched_in_backward"]
    events = list(red.events)
    red.wait()
    torch.cuda.synchronize()
    g_rccl = opt.flat_g.clone()
    done_before_end = sum(1 for _, ev in events if ev.elapsed_time(ev_bwd_end) > 0.0)
    in_wait = red.stats["launched_in_wait"]

    # (2) two optimizer steps through the normal path: pre-step hook joins the handles, then the Adam kernel
    for _ in range(2):
        fwd_bwd()
        opt.step()
    torch.cuda.synchronize()
    p_rccl = opt.flat_p.clone()
    obs = {"n_buckets": len(red.buckets), "launched_in_backward": in_backward, "launched_in_wait": in_wait,
           "buckets_complete_before_backward_end": done_before_end,
           "grad_bit_exact": bool(torch.equal(g_ref, g_rccl)), "grad_absmax": float(g_ref.abs().max()),
           "params_bit_exact": bool(torch.equal(p_ref, p_rccl)), "params_moved": bool(not torch.equal(p_ref, p0)),
           "steps_joined": red.stats["steps"], "backend": dist.get_backend(), "world": dist.get_world_size()}
    with open(out_path, "w") as f:
        json.dump(obs, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
