"""``@persistent_class`` decorator: records constructor arguments so a module can be re-created.

The reference's version (torch_utils/persistence.py:34-126) also embeds the defining module's
source text into pickles; that storage format is outside the G/D hot path
and is not reproduced. What layer code relies on -- ``init_args`` / ``init_kwargs`` attributes and
the decorator being transparent to ``isinstance`` and ``state_dict`` -- is kept.
"""

import copy

def persistent_class(orig_class):
    assert isinstance(orig_class, type)

    class Decorator(orig_class):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._init_args = copy.deepcopy(args)
            self._init_kwargs = copy.deepcopy(kwargs)

        @property
        def init_args(self):
            return copy.deepcopy(self._init_args)

        @property
        def init_kwargs(self):
            return copy.deepcopy(self._init_kwargs)

    Decorator.__name__ = orig_class.__name__
    Decorator.__qualname__ = orig_class.__qualname__
    Decorator.__module__ = orig_class.__module__
    Decorator.__doc__ = orig_class.__doc__
    return Decorator

def is_persistent(obj):
    return hasattr(obj, '_init_kwargs') or (isinstance(obj, type) and hasattr(obj, 'init_kwargs'))
