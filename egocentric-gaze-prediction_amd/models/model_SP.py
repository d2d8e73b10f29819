"""Mirror of the reference's ``models/model_SP.py``: two-stream saliency encoder-fusion-decoder (SP module).

Same constructor, attributes (features_s, features_t, relu, fusion, pool3d, bn, decoder, final),
state-dict keys (215 entries) and weight init as models/model_SP.py:5-65; ``forward`` runs the fused
HIP blocks of ``functions.py``.  A forward hook on ``features_s`` (AT.py:105) still fires and sees the
post-ReLU (B,512,14,14) activation.
"""
import torch.nn as nn
import torch.nn.modules.module as _mod

import torch

from ..functions import FusionBlock
from ..streams import fork
from ..utils import FusedSequential, init_like_reference

_STAGGER = False       # (round 2: the two encoders started one block apart so that they did not run in lock-step, +0.4 %.  Since the
                       # host issues them block by block, alternately, and the flow stack arrives re-laid out, the delay only costs:
                       # 30.12 / 30.17 / 30.13 -> 30.07 / 30.01 / 29.97 ms same box, round 6; test_interleaved_encoder_issue_* runs both)
_INTERLEAVE = True     # the host issues the two encoders block by block, alternately (A/B decided in round 5: 25.63 -> 25.38 ms; test_interleaved_encoders_* flips it)

# models/model_SP.py:13-31 as (Cin, Cout) 3x3+ReLU blocks and 'U' = nearest x2 upsample; a 1x1 head follows
_DECODER_PLAN = [(512, 512), (512, 512), 'U', (512, 512), (512, 512), (512, 512), 'U', (512, 256), (256, 256),
                 (256, 256), 'U', (256, 128), (128, 128), 'U', (128, 64), (64, 64)]


class model_SP(nn.Module):
    def __init__(self, features_s, features_t):
        super(model_SP, self).__init__()
        self.features_t = features_t
        self.features_s = features_s
        self.relu = nn.ReLU()
        self.fusion = nn.Conv3d(512, 512, kernel_size=(1, 3, 3), padding=(0, 1, 1))
        self.pool3d = nn.MaxPool3d(kernel_size=(2, 1, 1), padding=0)
        self.bn = nn.BatchNorm2d(512)
        layers = []
        for item in _DECODER_PLAN:
            if item == 'U':
                layers.append(nn.Upsample(scale_factor=2))
            else:
                layers += [nn.Conv2d(item[0], item[1], kernel_size=3, padding=1), nn.ReLU(inplace=True)]
        layers.append(nn.Conv2d(64, 1, kernel_size=1, padding=0))
        self.decoder = FusedSequential(*layers)            # 29 children, indices as the reference's
        self.final = nn.Sigmoid()
        self._initialize_weights()

    def _stack_buffer(self, x_s, x_t):
        """One (2B, h, w, 512) NHWC buffer whose halves the two encoders write directly: the reference's
        ``torch.cat((x_s, x_t), 2)`` (models/model_SP.py:38-39) then costs nothing.  None when the encoders are not the
        stock cfg['D'] stacks (their output geometry is then unknown here) -- FusionBlock copies in that case."""
        try:
            n_pool = sum(isinstance(m, nn.MaxPool2d) for m in self.features_s.children())
            last = [m for m in self.features_s.children() if isinstance(m, nn.Conv2d)][-1]
            last_t = [m for m in self.features_t.children() if isinstance(m, nn.Conv2d)][-1]
            ok = (isinstance(self.features_s, FusedSequential) and isinstance(self.features_t, FusedSequential)
                  and x_s.is_cuda and x_s.shape[0] == x_t.shape[0] and x_s.shape[2:] == x_t.shape[2:]
                  and last.out_channels == last_t.out_channels and isinstance(list(self.features_s.children())[-1], nn.ReLU)
                  and isinstance(list(self.features_t.children())[-1], nn.ReLU)
                  and sum(isinstance(m, nn.MaxPool2d) for m in self.features_t.children()) == n_pool)
        except Exception:
            ok = False
        if not ok:
            return None
        B, _, Hh, Ww = x_s.shape
        return torch.empty((2 * B, Hh >> n_pool, Ww >> n_pool, last.out_channels), dtype=torch.float32, device=x_s.device)

    def forward(self, x_s, x_t):
        stack = self._stack_buffer(x_s, x_t)
        B = x_s.shape[0]
        stagger = None
        with fork("encoder_t") as f:                     # the two encoders are independent: two HIP streams
            if f.enabled:
                x_t.record_stream(torch.cuda.current_stream())
                if stack is not None:
                    stack.record_stream(torch.cuda.current_stream())
                if _STAGGER:
                    stagger = torch.cuda.Event()
            # (the interleaved issue bypasses features_s.__call__ and fires its plain forward hooks by hand below; anything else
            # hooked onto either encoder -- pre-hooks, kwargs hooks, always-call hooks, GLOBAL module hooks -- takes the
            # sequential path, where __call__ runs them all: ADVICE r5)
            interleave = (_INTERLEAVE and f.enabled and isinstance(self.features_t, FusedSequential)
                          and isinstance(self.features_s, FusedSequential) and not self.features_t._forward_hooks
                          and not self.features_t._forward_pre_hooks and not self.features_s._forward_pre_hooks
                          and not getattr(self.features_s, "_forward_hooks_with_kwargs", None)
                          and not getattr(self.features_s, "_forward_hooks_always_called", None)
                          and not _mod._global_forward_hooks and not _mod._global_forward_pre_hooks)
            if interleave:
                gen_t = self.features_t.blocks(x_t, out_buf=stack[B:] if stack is not None else None,
                                               after_first_block=(lambda: stagger.record()) if stagger is not None else None)
                x_t = next(gen_t)                        # the flow encoder leads by one block
            else:
                x_t = self.features_t(x_t, out_buf=stack[B:] if stack is not None else None,
                                      after_first_block=(lambda: stagger.record()) if stagger is not None else None)
        if interleave:
            # Both encoders block by block, alternately: the host issues ~100 launches per encoder, and issuing one encoder after
            # the other left the second stream empty while the device ran the first (its BatchNorm passes with nothing beside
            # them).  Autograd replays the nodes in reverse creation order, so the backward pass alternates the same way.
            if stagger is not None:
                torch.cuda.current_stream().wait_event(stagger)
            x_in = x_s
            gen_s = self.features_s.blocks(x_s, out_buf=stack[:B] if stack is not None else None)
            live_s = live_t = True
            while live_s or live_t:
                if live_s:
                    try:
                        x_s = next(gen_s)
                    except StopIteration:
                        live_s = False
                if live_t:
                    with torch.cuda.stream(f.side):
                        try:
                            x_t = next(gen_t)
                        except StopIteration:
                            live_t = False
            for hook in self.features_s._forward_hooks.values():         # (AT.py:105 hooks features_s; __call__ was bypassed)
                r = hook(self.features_s, (x_in,), x_s)
                if r is not None:
                    x_s = r
            f.join(x_t)
            bn = self.bn
            nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
            x_fused = FusionBlock.apply(x_s, x_t, self.fusion.weight, self.fusion.bias, bn.weight, bn.bias,
                                        bn.running_mean, bn.running_var, bn.training, float(bn.momentum), float(bn.eps), nbt)
            return self.decoder(x_fused, fuse_sigmoid=True)
        if stagger is not None:
            # The encoders have the same layer sequence; started together they stay in lock-step (both in a conv, then both in
            # a BN pass) and the matrix cores idle during every BN pass.  Holding the RGB encoder back by the flow encoder's
            # first block puts them half a layer apart: one stream's HBM-bound pass runs under the other's MFMA-bound conv.
            torch.cuda.current_stream().wait_event(stagger)
        x_s = self.features_s(x_s, out_buf=stack[:B] if stack is not None else None)   # (B,512,h,w) channels_last; hooks fire here
        f.join(x_t)
        bn = self.bn
        nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None     # bumped by the stats kernel
        x_fused = FusionBlock.apply(x_s, x_t, self.fusion.weight, self.fusion.bias, bn.weight, bn.bias,
                                    bn.running_mean, bn.running_var, bn.training, float(bn.momentum), float(bn.eps), nbt)
        return self.decoder(x_fused, fuse_sigmoid=True)  # decoder + self.final fused

    def _initialize_weights(self):
        init_like_reference(self)
