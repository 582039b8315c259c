import functools
import inspect
from types import SimpleNamespace


class ConfigMixin:
    pass


def register_to_config(init):
    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        self.config = SimpleNamespace(**cfg)
    return wrapper
