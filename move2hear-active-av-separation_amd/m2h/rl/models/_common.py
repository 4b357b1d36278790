"""Shared helpers of the m2h drop-in modules."""
import torch


class PackedCache:
    """Derived device tensors (packed weights, folded BN) rebuilt when any source tensor changes in place or is replaced."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors)
        if key != self.key:
            self.val = build()
            self.key = key
        return self.val


def check_inference(module, *tensors):
    """The HIP path has no autograd yet: refuse (loudly) to run where the reference would have recorded a graph."""
    if torch.is_grad_enabled() and (any(p.requires_grad for p in module.parameters())
                                    or any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)):
        raise NotImplementedError(
            "m2h %s: backward through the HIP path is not built yet; call under torch.no_grad() "
            "(rollout / evaluation) or freeze the module" % type(module).__name__)
