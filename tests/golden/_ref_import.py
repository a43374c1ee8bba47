"""Loader used ONLY by tests/golden/make_golden.py in the build container.

It makes the reference's hot-path modules importable from /root/reference on
Python 3.10 by registering empty placeholder modules for import-time-only
dependencies that are not installed here (wandb, ray, natsort, rmet, dask,
implicit, mashumaro, param, ...) and a StrEnum backport.  None of those
packages take part in the arithmetic that the golden vectors pin: configs are
constructed directly as dataclasses and `rmet` (metric arithmetic) is never
called.  Nothing here ships to the GPU box and nothing of the reference is
copied: the reference stays at /root/reference and is imported from there.
"""
import enum
import sys
import types
import dataclasses

REF = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)

    if not hasattr(enum, 'StrEnum'):
        class StrEnum(str, enum.Enum):
            def _generate_next_value_(name, start, count, last_values):
                return name.lower()

            def __str__(self):
                return str(self.value)
        enum.StrEnum = StrEnum

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, item):
            return _Any()

    # param: descriptors collapse to their default value
    def _default(default=None, *a, **k):
        return default

    class Parameterized:
        pass
    _mod('param', Parameterized=Parameterized, Selector=_default, Integer=_default, Number=_default)

    # mashumaro: dataclasses are built directly, (de)serialisation is unused
    class DataClassDictMixin:
        def to_dict(self):
            return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}

        @classmethod
        def from_dict(cls, d):
            return cls(**d)

    class MissingField(Exception):
        pass

    class DataClassYAMLMixin(DataClassDictMixin):
        pass
    _mod('mashumaro', DataClassDictMixin=DataClassDictMixin,
         exceptions=_mod('mashumaro.exceptions', MissingField=MissingField))
    _mod('mashumaro.mixins')
    _mod('mashumaro.mixins.yaml', DataClassYAMLMixin=DataClassYAMLMixin)

    def _placeholder_getattr(item):
        if item.startswith('__'):
            raise AttributeError(item)
        return _Any()

    import torch  # noqa: F401  (must be imported before placeholders exist)
    import torch.utils.data.dataloader as dl
    import typing
    import scipy.sparse, sklearn.preprocessing, sklearn.manifold, pandas  # noqa: F401,E401

    for name in ['wandb', 'natsort', 'rmet', 'gdown', 'zenodopy', 'wikipedia', 'timeout_decorator',
                 'dask', 'dask.dataframe', 'implicit', 'implicit.als', 'ray', 'ray.air', 'ray.air.session',
                 'PIL', 'PIL.Image', 'matplotlib', 'matplotlib.pyplot']:
        if name in sys.modules:
            continue
        try:
            __import__(name)
        except Exception:
            m = _mod(name)
            m.__getattr__ = _placeholder_getattr
    sys.modules['ray.air'].session = sys.modules['ray.air.session']
    sys.modules['natsort'].natsorted = sorted

    if not hasattr(dl, 'T_co'):
        dl.T_co = typing.TypeVar('T_co', covariant=True)
    if not hasattr(dl, '_worker_init_fn_t'):
        dl._worker_init_fn_t = typing.Callable
