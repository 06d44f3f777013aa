"""Minimal ``dnnlib`` surface used on the hot path (reference: dnnlib/util.py:41-53 EasyDict,
:287-289 construct_class_by_name). The reference's logging / URL-cache helpers are not part of
the G/D forward+backward path and are not reproduced here."""

from .util import EasyDict, construct_class_by_name, get_obj_by_name, call_func_by_name
from . import util
