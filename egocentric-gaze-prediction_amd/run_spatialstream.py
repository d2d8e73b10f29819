"""Mirror of the reference's ``run_spatialstream.py`` (BASELINE config 1): single-image demo of the spatial stream --
``VGG`` (VGG16-BN encoder + a 14-conv decoder with three 512->512 convs before the first upsample,
run_spatialstream.py:17-53) -> gaze map -> centre of mass -> 3x3 crop of the conv5_3 map -> channel weights ->
weighted map -> bilinear x16 -> ``late_fusion(out, weighted)`` (:121-138).

The reference script parses argv and runs at import; here the same pieces are importable (``VGG``, ``crop_feature1``,
``get_weighted``, ``totensor``, ``toim``, ``predict``) and ``main()`` keeps the CLI (--trained_model, --trained_late,
--dir, --device).  Encoder, decoder, late fusion AND the glue between them run on HIP kernels (``SpatialPipeline``:
uint8 centre of mass, crop mean, weighted min-max map, bilinear x16 -- no host round trip, capturable as one hipGraph);
``predict`` keeps the reference's host-side glue (scipy centre of mass, torch slicing) as the comparison path.
"""
import argparse
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .models.late_fusion import late_fusion
from .utils import FusedSequential, cfg, init_like_reference, make_layers

_DECODER_PLAN = [(512, 512), (512, 512), (512, 512), 'U', (512, 512), (512, 512), (512, 512), 'U', (512, 256),
                 (256, 256), (256, 256), 'U', (256, 128), (128, 128), 'U', (128, 64), (64, 64)]


class VGG(nn.Module):
    def __init__(self, features):
        super(VGG, self).__init__()
        self.features = features
        for param in self.features.parameters():
            param.requires_grad = False
        layers = []
        for item in _DECODER_PLAN:
            if item == 'U':
                layers.append(nn.Upsample(scale_factor=2))
            else:
                layers += [nn.Conv2d(item[0], item[1], kernel_size=3, padding=1), nn.ReLU(inplace=True)]
        layers.append(nn.Conv2d(64, 1, kernel_size=1, padding=0))
        self.decoder = FusedSequential(*layers)              # 31 children, indices as the reference's
        self.final = nn.Sigmoid()
        init_like_reference(self)

    def forward(self, x):
        xo = self.features(x)
        return self.decoder(xo, fuse_sigmoid=True), xo


def crop_feature1(feature, maxind, size):
    """size x size window of the conv5_3 map around gaze_point // 16 (run_spatialstream.py:85-94)."""
    H = feature.size(2)
    lo, hi = size // 2, int(math.ceil(size / 2.0))
    f = np.clip(np.array(maxind) // 16, lo, H - hi)
    return feature[:, :, int(f[0] - lo):int(f[0] + hi), int(f[1] - lo):int(f[1] + hi)]


def get_weighted(chn_weight, feature):
    feature = torch.sum(feature * chn_weight.view(1, 512, 1, 1), 1)
    feature = feature - torch.min(feature)
    return (feature / torch.max(feature)).unsqueeze(0)


def totensor(im):
    """BGR uint8 (H,W,3) -> normalised (1,3,H,W); ImageNet constants applied in BGR order like the reference."""
    t = torch.from_numpy(im.transpose((2, 0, 1)).copy()).float().div(255)
    t = (t - torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    return t.unsqueeze(0)


def toim(ten):
    return (ten.squeeze().cpu().numpy() * 255).astype(np.uint8)


class SpatialPipeline(nn.Module):
    """The whole loop body of run_spatialstream.py:125-138 after the image upload as ONE device-resident forward:
    VGG -> uint8 centre of mass -> 3 x 3 crop mean of conv5_3 -> channel-weighted, min-max normalised map -> bilinear x16 ->
    ``late_fusion(out, weighted)``.  Every stage is a HIP kernel of this package on the current stream -- no device-to-host
    copy, no scipy, no tensor-library kernel (the reference does a D2H copy + ndimage.center_of_mass + torch slicing / mean /
    upsample per frame) -- so ``graphs.GraphedModule(SpatialPipeline(model, lf), ...)`` replays a frame as one hipGraph.
    Returns (out, feat, com, vec, weighted, fin); com = (1, 2) float64 (row, col) on the device."""

    def __init__(self, model, lf, crop_size=3):
        super().__init__()
        self.model, self.lf, self.crop_size = model, lf, crop_size

    def forward(self, im):
        from . import hipops as H
        from .functions import to_nhwc
        out, feat = self.model(im)                                   # (B,1,224,224), (B,512,14,14) channels_last
        B, _, Hh, Ww = out.shape
        com, gp = H.u8_center_of_mass(out.reshape(B, Hh, Ww))        # `toim` + ndimage.center_of_mass
        fn = to_nhwc(feat)
        vec = H.crop_mean(fn, gp, self.crop_size, Hh // fn.shape[1])  # crop_feature1 + mean -> chn_weight (B, 512)
        wmap = H.weighted_minmax(fn, vec)                            # get_weighted -> (B, 14, 14)
        fused = torch.empty((B, 2, Hh, Ww), dtype=torch.float32, device=out.device)     # late_fusion's cat((f, g), 1)
        for b in range(B):
            H.copy_into(fused[b, 0], out[b, 0])
        H.bilinear_up(wmap, Hh // wmap.shape[1], align_corners=False, out=fused[:, 1])
        fin = self.lf.fusion(fused, fuse_sigmoid=True)               # late_fusion.forward without the cat kernel
        return out, feat, com, vec, fused[:, 1:2], fin


def predict_device(pipeline, im_bgr_u8, device):
    """`predict` with the glue on the device: ``pipeline`` = SpatialPipeline or a GraphedModule of one."""
    im = totensor(im_bgr_u8).to(device)
    with torch.no_grad():
        out, feat, com, vec, weighted, fin = pipeline(im)
    return {"out": out, "feat": feat, "imq": toim(out), "predicted": com[0].cpu().numpy(), "vec": vec[0],
            "weighted": weighted, "fin": fin}


def predict(model, lf, im_bgr_u8, device, graphed=None):
    """One iteration of the reference's loop body (run_spatialstream.py:123-139). Returns a dict of stages.
    ``graphed``: a graphs.GraphedModule of ``model`` (one hipGraph replay instead of ~100 launches for the batch-1 forward;
    its outputs are static buffers, valid until the next call)."""
    from scipy import ndimage
    im = totensor(im_bgr_u8).to(device)
    with torch.no_grad():
        out, feat = graphed(im) if graphed is not None else model(im)
    imq = toim(out)
    predicted = ndimage.center_of_mass(imq)
    vec = crop_feature1(feat, predicted, 3)
    vec = vec.contiguous().view(vec.size(0), vec.size(1), -1)
    vec = torch.mean(vec, 2).squeeze()
    weighted = get_weighted(vec, feat)
    weighted = torch.nn.functional.interpolate(weighted.contiguous(), scale_factor=16, mode='bilinear')
    with torch.no_grad():
        fin = lf(out, weighted)                      # NB (SP map, AT map): reverse of LF.py:90
    return {"out": out, "feat": feat, "imq": imq, "predicted": np.array(predicted), "vec": vec,
            "weighted": weighted, "fin": fin}


def main(argv=None):
    from .data._io import imread, imwrite, resize
    p = argparse.ArgumentParser()
    p.add_argument('--trained_model', default='models/spatial.pth.tar', required=False)
    p.add_argument('--trained_late', default='models/late.pth.tar', required=False)
    p.add_argument('--dir', required=True)
    p.add_argument('--device', default='0', help='GPU index')
    p.add_argument('--hipgraph', action='store_true', help='replay the whole per-frame pipeline as one captured hipGraph')
    args = p.parse_args(argv)
    device = torch.device('cuda:' + args.device)
    model = VGG(make_layers(cfg['D'], 3))
    model.load_state_dict(torch.load(args.trained_model, map_location='cpu', weights_only=False)['state_dict'])
    model.to(device).eval()
    lf = late_fusion()
    lf.load_state_dict(torch.load(args.trained_late, map_location='cpu', weights_only=False)['state_dict'])
    lf.to(device).eval()
    pipeline = SpatialPipeline(model, lf).eval()
    if args.hipgraph:
        from .graphs import GraphedModule
        pipeline = GraphedModule(pipeline, (torch.zeros(1, 3, 224, 224, device=device),))
    for imname in [k for k in os.listdir(args.dir) if 'img' in k]:
        im0 = imread(os.path.join(args.dir, imname))
        res = predict_device(pipeline, resize(im0, (224, 224)), device)
        fin = resize(toim(res["fin"]), (im0.shape[1], im0.shape[0]))
        try:
            import cv2
            overlay = im0 * 0.7 + cv2.applyColorMap(fin, cv2.COLORMAP_JET) * 0.3
        except ImportError:
            overlay = im0 * 0.7 + np.repeat(fin[:, :, None], 3, 2) * 0.3
        imwrite(os.path.join(args.dir, 'out_' + imname[3:]), overlay.astype(np.uint8))
        print('result saved to ' + os.path.join(args.dir, 'out_' + imname[3:]))


if __name__ == '__main__':
    main()
