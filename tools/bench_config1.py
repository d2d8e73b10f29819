"""Per-frame time of BASELINE config 1 (run_spatialstream.py:123-139) on the HIP path: host glue (`predict`: D2H + scipy centre
of mass + torch slicing / mean / bilinear, VGG forward eager or graphed) vs the device-resident pipeline (`SpatialPipeline`)
eager and as ONE captured hipGraph.  Usage: python tools/bench_config1.py [--frames N]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egaze_amd  # noqa
from egaze_amd.graphs import GraphedModule
from egaze_amd.models.late_fusion import late_fusion
from egaze_amd.run_spatialstream import VGG, SpatialPipeline, predict, predict_device
from egaze_amd.utils import cfg, make_layers


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = VGG(make_layers(cfg['D'], 3)).to(dev).eval()
    lf = late_fusion().to(dev).eval()
    ims = [np.random.RandomState(i).randint(0, 256, (224, 224, 3)).astype(np.uint8) for i in range(8)]
    pipe = SpatialPipeline(model, lf).eval()
    gm = GraphedModule(model, (torch.zeros(1, 3, 224, 224, device=dev),))
    gp = GraphedModule(pipe, (torch.zeros(1, 3, 224, 224, device=dev),))
    torch.set_num_threads(8)             # the host-side `totensor` is five small torch-CPU ops: 256 threads make each one slow
    legs = [("host glue, eager VGG (round 2 default)", lambda im: predict(model, lf, im, dev)),
            ("host glue, graphed VGG (round 2 --hipgraph)", lambda im: predict(model, lf, im, dev, gm)),
            ("device glue, eager", lambda im: predict_device(pipe, im, dev)),
            ("device glue, ONE hipGraph per frame", lambda im: predict_device(gp, im, dev))]
    for name, fn in legs:
        for i in range(5):
            fn(ims[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.frames):
            r = fn(ims[i % 8])
            r["fin"].cpu()                       # the script writes the fused map of every frame (toim -> imwrite)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.frames
        print(f"{name:48s} {dt * 1e3:7.3f} ms / frame  {1 / dt:7.0f} frames/s   (host totensor + upload + toim included)", flush=True)
    # the device part alone: normalised frame already resident, only the fused map is read back
    from egaze_amd.run_spatialstream import totensor
    dims = [totensor(im).to(dev) for im in ims]
    from scipy import ndimage

    def host_glue(im, g):
        with torch.no_grad():
            out, feat = g(im) if g is not None else model(im)
        imq = (out.squeeze().cpu().numpy() * 255).astype(np.uint8)
        from egaze_amd.run_spatialstream import crop_feature1, get_weighted
        vec = crop_feature1(feat, ndimage.center_of_mass(imq), 3)
        vec = torch.mean(vec.contiguous().view(vec.size(0), vec.size(1), -1), 2).squeeze()
        w = torch.nn.functional.interpolate(get_weighted(vec, feat).contiguous(), scale_factor=16, mode='bilinear')
        with torch.no_grad():
            return lf(out, w)

    def dev_glue(im, p):
        with torch.no_grad():
            return p(im)[5]
    legs = [("host glue, eager VGG", lambda im: host_glue(im, None)), ("host glue, graphed VGG", lambda im: host_glue(im, gm)),
            ("device glue, eager", lambda im: dev_glue(im, pipe)), ("device glue, ONE hipGraph per frame", lambda im: dev_glue(im, gp))]
    for name, fn in legs:
        for i in range(5):
            fn(dims[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.frames):
            fn(dims[i % 8]).cpu()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.frames
        print(f"{name:48s} {dt * 1e3:7.3f} ms / frame  {1 / dt:7.0f} frames/s   (frame resident in HBM, fused map read back)", flush=True)


if __name__ == "__main__":
    main()
