"""``@persistent_class``: modules that remember their constructor arguments and pickle in the reference's snapshot format.

Stands where torch_utils/persistence.py of the reference stands (decorator :34-126, ``is_persistent`` :130-140,
``import_hook`` :144-173, ``_reconstruct_persistent_obj`` :177-200). The pickle layout is the reference's, so
snapshots interchange: a persistent object reduces to ``(_reconstruct_persistent_obj, (meta,), None)`` with
``meta = dict(type='class', version, module_src, class_name, state)``; ``module_src`` is the source text of the
defining module and ``state`` the object's ``__dict__``.

Loading differs from the reference on purpose. The reference executes ``module_src``; for a snapshot written by the
reference that text is its own ``training/networks.py``, which does not import on ROCm and would bind the
model to the reference's operators. Here a pickled class is re-bound BY NAME to the class of that name in a module
that registered persistent classes with this process (``training.networks`` of this package): the object is rebuilt
through the local constructor from the recorded ``init_args`` / ``init_kwargs`` and then takes the pickled
parameters, buffers and sub-modules. Only when no local class has that name is the embedded source executed, as the
reference does. ``import_hook`` keeps the reference's contract (called with ``meta``, returns ``meta``).
"""

import copy
import inspect
import sys
import types
import uuid

import dnnlib

_version = 6                    # pickle format version of the reference (persistence.py:28)
_decorators = set()             # {decorator_class, ...}
_import_hooks = []              # [hook_function, ...]
_local_classes = {}             # class name -> local persistent (decorated) class, filled by persistent_class()

rebind_to_local_classes = True  # False: always execute the embedded source (the reference's behaviour)

#----------------------------------------------------------------------------

def persistent_class(orig_class):
    """Class decorator: records constructor arguments (``init_args``, ``init_kwargs``) and makes instances pickle
    with their module's source text (reference :34-126)."""
    assert isinstance(orig_class, type)
    if is_persistent(orig_class):
        return orig_class
    assert orig_class.__module__ in sys.modules
    orig_module = sys.modules[orig_class.__module__]

    class Decorator(orig_class):
        _orig_class_name = orig_class.__name__

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._init_args = copy.deepcopy(args)
            self._init_kwargs = copy.deepcopy(kwargs)

        @property
        def init_args(self):
            return copy.deepcopy(self._init_args)

        @property
        def init_kwargs(self):
            return dnnlib.EasyDict(copy.deepcopy(self._init_kwargs))

        def __reduce__(self):
            # (callable, args, state) of the base class, re-targeted at the snapshot format unless a base class
            # already did so
            ctor, ctor_args, state, *rest = list(super().__reduce__()) + [None, None]
            if ctor is _reconstruct_persistent_obj:
                return (ctor, ctor_args, state)
            meta = dict(type='class', version=_version, module_src=_module_to_src(orig_module),
                        class_name=self._orig_class_name, state=state)
            return (_reconstruct_persistent_obj, (meta,), None)

    Decorator.__name__ = orig_class.__name__
    Decorator.__qualname__ = orig_class.__qualname__
    Decorator.__module__ = orig_class.__module__
    Decorator.__doc__ = orig_class.__doc__
    _decorators.add(Decorator)
    if not orig_module.__name__.startswith('_imported_module_'):        # classes of executed snapshot source stay anonymous
        _local_classes.setdefault(orig_class.__name__, Decorator)
    return Decorator

#----------------------------------------------------------------------------

def is_persistent(obj):
    """Is ``obj`` a persistent class or an instance of one? (reference :130-140)"""
    try:
        if obj in _decorators:
            return True
    except TypeError:
        pass
    return type(obj) in _decorators

#----------------------------------------------------------------------------

def import_hook(hook):
    """Register ``hook(meta) -> meta``, called for every persistent object being unpickled; ``meta`` is an
    ``EasyDict`` with ``type``, ``version``, ``module_src``, ``class_name``, ``state`` (reference :144-173)."""
    assert callable(hook)
    _import_hooks.append(hook)
    return hook

#----------------------------------------------------------------------------

def _reconstruct_persistent_obj(meta):
    """Called by ``pickle`` to rebuild a persistent object (reference :177-200; see the module docstring for the
    name-based re-binding)."""
    meta = dnnlib.EasyDict(meta)
    meta.state = dnnlib.EasyDict(meta.state)
    for hook in _import_hooks:
        meta = hook(meta)
        assert meta is not None
    assert meta.version == _version, f'persistent pickle of format {meta.version}, this loader reads {_version}'
    assert meta.type == 'class'

    decorator_class = _local_classes.get(meta.class_name) if rebind_to_local_classes else None
    if decorator_class is not None:
        obj = decorator_class.__new__(decorator_class)
        args, kwargs = meta.state.get('_init_args', ()), meta.state.get('_init_kwargs', {})
        decorator_class.__init__(obj, *args, **kwargs)          # local structure and derived attributes
        _adopt_state(obj, meta.state)
        return obj

    module = _src_to_module(meta.module_src)                    # foreign class without a local namesake
    decorator_class = persistent_class(module.__dict__[meta.class_name])
    obj = decorator_class.__new__(decorator_class)
    setstate = getattr(obj, '__setstate__', None)
    if callable(setstate):
        setstate(meta.state)
    else:
        obj.__dict__.update(meta.state)
    return obj

def _adopt_state(obj, state):
    """Take parameters, buffers, sub-modules and the training flag of a pickled ``nn.Module`` ``__dict__``; every other
    attribute keeps the value the local constructor gave it."""
    params, buffers, modules = state.get('_parameters'), state.get('_buffers'), state.get('_modules')
    if params is None:                                         # not an nn.Module: plain attribute state
        obj.__dict__.update(state)
        return
    for name, value in params.items():
        if name not in obj._parameters:
            raise KeyError(f'{type(obj).__name__}: pickled parameter {name!r} does not exist in the local class')
        if value is not None and obj._parameters[name] is not None and value.shape != obj._parameters[name].shape:
            raise ValueError(f'{type(obj).__name__}.{name}: pickled shape {tuple(value.shape)} != {tuple(obj._parameters[name].shape)}')
        obj._parameters[name] = value
    for name, value in (buffers or {}).items():
        if name not in obj._buffers:
            raise KeyError(f'{type(obj).__name__}: pickled buffer {name!r} does not exist in the local class')
        obj._buffers[name] = value
    for name, value in (modules or {}).items():
        if name not in obj._modules:
            raise KeyError(f'{type(obj).__name__}: pickled sub-module {name!r} does not exist in the local class')
        obj._modules[name] = value
    if 'training' in state:
        obj.training = state['training']

#----------------------------------------------------------------------------

class _Sources:
    """Two-way map between modules and their source text.  A text that belongs to no known module becomes an
    anonymous module the first time it is asked for (that is how a foreign snapshot's classes come to life when no
    local class takes their place)."""
    def __init__(self):
        self.by_module, self.by_text = {}, {}

    def text_of(self, module):
        if module not in self.by_module:
            text = inspect.getsource(module)
            self.by_module[module], self.by_text[text] = text, module
        return self.by_module[module]

    def module_of(self, text):
        if text not in self.by_text:
            module = types.ModuleType('_imported_module_' + uuid.uuid4().hex)
            sys.modules[module.__name__] = module
            self.by_module[module], self.by_text[text] = text, module
            exec(text, module.__dict__)  # pylint: disable=exec-used
        return self.by_text[text]

_sources = _Sources()
_module_to_src = _sources.text_of
_src_to_module = _sources.module_of

#----------------------------------------------------------------------------
