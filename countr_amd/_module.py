"""Plumbing shared by the drop-in nn.Modules: parameters alias the engine's flat fp32 master buffer; the engine's
low-precision / permuted shadows are refreshed whenever a parameter's version counter moved."""
import torch
import torch.nn as nn


class HipModule(nn.Module):
    _eng = None
    _versions = None

    def _make_engine(self, shapes, device):          # -> Engine
        raise NotImplementedError

    def _is_trainable(self, name):
        raise NotImplementedError

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._eng = None  # parameters moved / re-typed: repack on the next forward
        return out

    def _engine(self):
        params = list(self.named_parameters())
        dev = params[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("%s (HIP engine) needs its parameters on a GPU: call model.to('cuda'); "
                               "there is no CPU fallback" % type(self).__name__)
        if self._eng is None:
            shapes = [(n, tuple(p.shape)) for n, p in params]
            eng = self._make_engine(shapes, dev)
            with torch.no_grad():
                for n, p in params:
                    view = eng.pview(n)
                    view.copy_(p.data.float())
                    p.data = view  # parameters now alias the flat fp32 master buffer
            self._train_names = [n for n, _ in params if self._is_trainable(n)]
            self._train_params = [p for n, p in params if self._is_trainable(n)]
            self._eng = eng
            self._versions = None
        eng = self._eng
        # repack if someone replaced a parameter's storage (e.g. load_state_dict(assign=True))
        for n, p in params:
            if p.data_ptr() != eng.pview(n).data_ptr():
                with torch.no_grad():
                    eng.pview(n).copy_(p.data.float())
                    p.data = eng.pview(n)
                self._versions = None
        vers = self._version_sums(params)
        if vers != self._versions:  # optimizer.step() / load_state_dict changed values: refresh shadows
            # a torch optimizer stepping the trainable parameters through autograd moves only THEIR version counters: the frozen part's
            # shadows -- and the LayerNorm-fold / pre-scaled-q repack of the frozen encoder, ~36 torch matmuls -- are left alone then
            frozen_same = self._versions is not None and vers[0] == self._versions[0]
            eng.sync_weights(trainable_only=frozen_same)
            self._versions = vers
        return eng

    def _version_sums(self, params):
        """(sum of the frozen parameters' version counters, sum of the trainable ones')."""
        fr = tr = 0
        for n, p in params:
            if self._is_trainable(n):
                tr += p._version
            else:
                fr += p._version
        return (fr, tr)

    def mark_weights_synced(self):
        """The engine's fused AdamW keeps the shadows coherent itself."""
        self._versions = self._version_sums(list(self.named_parameters()))
