"""Packed-weights cache: nn.Linear tensors -> the kernels' MFMA fragment stream.

The packed buffer is rebuilt whenever any parameter's storage pointer or version
counter changes (optimizer.step, load_state_dict, .to(device)).  Edits made THROUGH ``param.data``
(``p.data.mul_()``, ``p.data.copy_()``: EMA / manual-init idioms) bump no version counter: call
``module.invalidate_packed()`` (NeRF / NoF) after them.

The cache holds ctypes structures with raw device pointers, which must not travel: pickling
(``torch.save(model)``) and ``copy.deepcopy(model)`` drop it and the copy re-packs on first use."""
import torch

from . import _lib as L


def _collect_params(m, out):
    for p in m._parameters.values():
        if p is not None:
            out.append(p)
    for c in m._modules.values():
        if c is not None:
            _collect_params(c, out)


class PackedWeights:
    def __init__(self):
        self.key = None
        self.buf = None
        self.desc = None
        self.keep = None   # contiguous fp32 views the descriptor points into

    def invalidate(self):
        self.key = self.buf = self.desc = self.keep = None

    def __getstate__(self):          # torch.save(module) / pickle: nothing cached travels
        return {}

    def __setstate__(self, state):
        self.invalidate()

    def __deepcopy__(self, memo):    # copy.deepcopy(module): the copy packs its own parameters
        return PackedWeights()

    def get(self, module, build_desc, bytes_fn, pack_fn, what, precision=0):
        # (module.parameters() builds names and de-duplicates through sets on every call: ~50 us per network and render
        #  pass, a tenth of a bf16 pass; this plain walk of the module tree reads the same tensors in the same order in ~10 us
        #  and still sees replaced parameters and sub-modules)
        params = []
        _collect_params(module, params)
        if not params:
            raise RuntimeError(f"{what}: module has no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError(f"moco_flow_amd.{what}: parameters are on '{dev}'. This is the MI355X (HIP) "
                               "path; there is no CPU implementation. Call .to('cuda') first.")
        key = (precision,) + tuple((p.data_ptr(), p._version, p.dtype) for p in params)
        if key != self.key:
            desc, keep = build_desc()
            nbytes = bytes_fn(desc, precision)
            if nbytes <= 0:
                L.check(-3, what)
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                L.check(pack_fn(desc, precision, buf.data_ptr(), L.current_stream(dev)), what + " pack")
            self.key, self.buf, self.desc, self.keep = key, buf, desc, keep
        return self.desc, self.buf
