"""hipGraph capture of a launch-bound forward pass (instead of a tracing compiler).

At small batch the eval-mode SP forward is ~150 short kernels: the device waits for the host to issue them one C-ABI call
at a time.  Every kernel of this package is launched on torch's current stream, so the whole forward -- including the
second encoder's side stream, which forks from and joins back into the capturing stream (streams.fork) -- can be captured
once into a hipGraph and replayed with a single launch.  Inputs are copied into static buffers, the outputs are the
captured tensors (valid until the next replay).  Only for no-grad forwards of modules whose parameters do not change
between replays (the packed weights are baked into the graph as pointers, their contents are read at replay time -- a
repack after an optimizer step would be missed, so GraphedModule checks the package's weight epoch and re-captures).
"""
from typing import Sequence

import torch

from . import hipops as H


class GraphedModule:
    def __init__(self, module: torch.nn.Module, example_inputs: Sequence[torch.Tensor], warmup: int = 2):
        self.module = module
        self._params = list(module.parameters()) + list(module.buffers())      # BN running statistics feed cached coefficients
        self.static_in = [t.clone() for t in example_inputs]
        self.warmup = warmup
        self.graph = None
        self.static_out = None
        self.epoch = None

    def _signature(self):
        """Changes whenever a parameter was rewritten (torch version counter, the fused optimizer's per-parameter epoch, a
        wholesale overwrite) or re-allocated: the packed weights inside the captured graph would then be stale."""
        v = e = 0
        for p in self._params:
            v += p._version
            e += getattr(p, "_egz_epoch", 0)
        return (H._WEIGHT_EPOCH[0], v, e, self._params[0].data_ptr() if self._params else 0)

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(self.warmup):          # lazy packings, workspaces and helper streams exist before the capture
                self.module(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), H.capture(self.graph):
            self.static_out = self.module(*self.static_in)
        self.epoch = self._signature()

    def __call__(self, *inputs):
        if self.module.training:
            raise RuntimeError("GraphedModule replays an eval-mode forward; call module.eval() first")
        if self.graph is None or self.epoch != self._signature():
            self._capture()
        for s, t in zip(self.static_in, inputs):
            if s.shape != t.shape:
                raise RuntimeError(f"GraphedModule was captured for inputs of shape {tuple(s.shape)}, got {tuple(t.shape)}")
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.static_out


class GraphedTrainStep:
    """One whole training step -- forward, loss, ``zero_grad``, backward, fused Adam -- captured into a hipGraph and replayed
    (the pattern of AT._GraphedSampleStep for a batched model): for steps whose device time is of the order of the host time
    to issue them (LF.trainLate at batch 32: ~110 launches, 1.0 ms of host work for 1.4 ms of kernels).

    ``forward_loss(*static_inputs) -> (loss, *outputs)`` runs the model and the criterion on the static input buffers; the
    optimizer must be a FusedAdam (its step counter moves to the device).  Inside the capture every weight-gradient fork joins
    back (functions._close_fork), BatchNorm running statistics and ``num_batches_tracked`` are updated by kernels, so a replay
    is exactly one eager step.  The first ``warm`` calls run eagerly (real steps: lazy packings, workspaces and helper
    streams come into being), the next one is captured.

    For single-chain models (late_fusion, lstmnet).  The two-stream SP step is NOT capturable this way -- its encoder streams and
    detached weight-gradient forks do not all rejoin the capturing stream inside the step (ending such a capture crashes inside
    the HIP runtime) -- and has nothing to gain: at batch 32 the host issues its ~700 launches in a third of the device time (§4)."""

    def __init__(self, forward_loss, optimizer, example_inputs: Sequence[torch.Tensor], warm: int = 2):
        self.fn, self.opt, self.warm = forward_loss, optimizer, warm
        self.static_in = [t.clone() for t in example_inputs]
        self.graph, self.calls, self.out = None, 0, None
        self.one = torch.ones((), device=self.static_in[0].device)
        self.opt.set_capturable(True)

    def _unit(self):
        res = self.fn(*self.static_in)
        loss = res[0]
        self.opt.zero_grad()
        loss.backward(gradient=self.one)
        self.opt.step()
        self.out = tuple(r.detach() for r in res)

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.calls += 1
        if self.graph is None and self.calls <= self.warm:
            self._unit()
        elif self.graph is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            count = self.opt.step_count
            with H.capture(g):
                self._unit()
            self.opt.step_count = count          # the capture ran the host side of step() without executing anything
            self.graph = g
            self._versions = [p._version for p in self.opt.params]
            g.replay()
            self.opt.note_replays(1)
        else:
            # the captured step rebuilds the packed weights AFTER its Adam update (FusedAdam.step -> H.refresh_packings), so its
            # forward pass trusts them: parameters overwritten from outside between two replays (load_state_dict, copy_)
            # must be repacked here
            versions = [p._version for p in self.opt.params]
            if versions != self._versions:
                H.refresh_packings(self.opt.params, force=True)
                self._versions = versions
            self.graph.replay()
            self.opt.note_replays(1)
        return self.out

    def close(self):
        self.opt.set_capturable(False)
