"""Worker of tests/test_hip_dp.py: ONE rank of a 2-rank data-parallel run of a real model of the path (split-half kernels,
HIP streams on, FusedAdam + dp.GradReducer) -- both ranks share GPU 0 and exchange gradients over gloo, the recipe for
exercising the N>1 path on a 1-GPU box.  argv: out-prefix size batch [kind]; kind = sp (model_SP + floss, BASELINE config 3;
default), at (lstmnet over T = size steps of 512-vectors + MSE, config 4) or lf (late_fusion + floss, config 5's last stage).
Launched by torch.distributed.run; writes its observations to argv[1].<rank>."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_prefix = sys.argv[1]
    size, batch = int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import egaze_amd  # noqa: F401
    import egaze_amd.hipops as H
    from egaze_amd import dp, streams, synthetic
    from egaze_amd.floss import floss
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.optim import FusedAdam
    from egaze_amd.utils import cfg, make_layers

    assert H.PRECISION == "split" and streams.ENABLED, "the test is about the default (split-half, streams on) path"
    kind = sys.argv[4] if len(sys.argv) > 4 else "sp"
    torch.manual_seed(1234 + rank)                       # replicas start DIFFERENT: attach() must broadcast rank 0's
    bucket_bytes = 8 * 1024 * 1024
    if kind == "sp":
        model = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev)
        model.train()
        crit = floss().to(dev)
        opt = FusedAdam(model.parameters(), lr=1e-4)
        b = synthetic.sp_batch(batch, size, dev, seed=100 + rank)

        def fwd_bwd():
            opt.zero_grad()
            out = model(b["image"], b["flow"])
            loss = crit(out, b["gt"].view(out.size()))
            loss.backward()
            return loss
    elif kind == "at":
        from egaze_amd.functions import MSELoss
        from egaze_amd.models.LSTMnet import lstmnet
        model = lstmnet().to(dev)
        model.train()
        opt = FusedAdam(model.parameters(), lr=1e-4)
        ab = synthetic.at_batch(size, batch, dev, seed=200 + rank)        # (T, B, 512) inputs and targets
        tgt = torch.tanh(ab["gt"])
        h0 = torch.zeros(2, batch, 512, device=dev)
        c0 = torch.zeros(2, batch, 512, device=dev)
        bucket_bytes = 4 * 1024 * 1024

        def fwd_bwd():
            opt.zero_grad()
            pred, _ = model(ab["input"], (h0, c0))
            loss = MSELoss.apply(pred, tgt)
            loss.backward()
            return loss
    else:
        from egaze_amd.models.late_fusion import late_fusion
        model = late_fusion().to(dev)
        model.train()
        crit = floss().to(dev)
        opt = FusedAdam(model.parameters(), lr=1e-4)
        g = torch.Generator().manual_seed(300 + rank)
        maps = [torch.rand(batch, 1, size, size, generator=g).to(dev) for _ in range(3)]
        bucket_bytes = 4 * 1024

        def fwd_bwd():
            opt.zero_grad()
            out = model(maps[0], maps[1])
            loss = crit(out, maps[2])
            loss.backward()
            return loss

    # (0) make the replicas identical first (what dp.attach does), then take the LOCAL gradient without any reducer
    dist_p = opt.flat_p.detach().cpu()
    dist.broadcast(dist_p, src=0)
    opt.flat_p.copy_(dist_p)
    H.bump_weight_epoch()
    fwd_bwd()
    streams.join_all_into_current()
    torch.cuda.synchronize()
    g_local = opt.flat_g.detach().cpu().clone()
    gathered = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(gathered, g_local)

    # (1) the same backward with the reducer attached: bucketed async all-reduce launched from the hooks
    red = dp.attach(opt, bucket_bytes=bucket_bytes)
    assert len(red.buckets) >= (4 if kind == "sp" else 2), len(red.buckets)
    fwd_bwd()
    red.wait()
    torch.cuda.synchronize()
    g_sum = opt.flat_g.detach().cpu().clone()

    # (2) two optimizer steps through the normal path (pre-step hook joins the collectives, 1/world inside Adam)
    losses = []
    for _ in range(2):
        loss = fwd_bwd()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    flat_p_2 = opt.flat_p.detach().cpu().clone()

    torch.save({"rank": rank, "g_local": gathered, "g_sum": g_sum, "flat_p": flat_p_2,
                "losses": losses, "n_buckets": len(red.buckets), "grad_scale": opt.grad_scale,
                "layout": [(n, o, p.numel()) for (n, p), o in zip(model.named_parameters(), opt.offsets)],
                "buckets": [tuple(b) for b in red.buckets]},
               f"{out_prefix}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
