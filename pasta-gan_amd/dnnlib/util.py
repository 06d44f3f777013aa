"""Attribute-style dict and by-name construction (reference: dnnlib/util.py:41-53, 228-289)."""

import importlib
from typing import Any


class EasyDict(dict):
    """dict whose items are also attributes: ``d.key`` is ``d['key']``."""

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        del self[name]


def get_obj_by_name(name: str) -> Any:
    """Resolve ``'package.module.attr[.attr...]'`` to the Python object it names."""
    parts = name.split('.')
    for split in range(len(parts) - 1, 0, -1):
        try:
            obj = importlib.import_module('.'.join(parts[:split]))
        except ImportError:
            continue
        try:
            for attr in parts[split:]:
                obj = getattr(obj, attr)
            return obj
        except AttributeError:
            continue
    raise ImportError(name)


def call_func_by_name(*args, func_name: str = None, **kwargs) -> Any:
    assert func_name is not None
    func = get_obj_by_name(func_name)
    assert callable(func)
    return func(*args, **kwargs)


def construct_class_by_name(*args, class_name: str = None, **kwargs) -> Any:
    """``construct_class_by_name(class_name='training.networks.Discriminator', **kw)``."""
    return call_func_by_name(*args, func_name=class_name, **kwargs)
