"""paddle.nn.initializer stand-ins.  Values are irrelevant (tests load explicit state dicts)."""
import torch


class _Init:
    def __init__(self, *a, **k):
        self.a, self.k = a, k

    def __call__(self, p):
        return p


class Constant(_Init):
    def __call__(self, p):
        with torch.no_grad():
            p.fill_(self.k.get("value", self.a[0] if self.a else 0.0))
        return p


class Assign(_Init):
    def __call__(self, p):
        with torch.no_grad():
            p.copy_(torch.as_tensor(self.a[0]).reshape(p.shape).to(p.dtype))
        return p


class XavierUniform(_Init):
    pass


class XavierNormal(_Init):
    pass


class KaimingUniform(_Init):
    pass


class KaimingNormal(_Init):
    pass


class Normal(_Init):
    pass


class Uniform(_Init):
    pass


def set_global_initializer(*a, **k):
    return None
