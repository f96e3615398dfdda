"""Minimal stand-in for the `params_proto` package (absent from this image).

Only the surface the Go1 scripts use: class-attribute config trees declared as
`class X(PrefixProto, cli=False)`, arbitrary attribute assignment on the classes, and
`vars(X)` returning a plain dict of the public parameters (nested trees become dicts).
"""


def _public(d):
    return {k: v for k, v in d.items()
            if not k.startswith("_") and not isinstance(v, (classmethod, staticmethod, property))
            and not (callable(v) and not isinstance(v, type))}


class Meta(type):
    def __new__(mcs, name, bases, ns, **kwargs):
        return super().__new__(mcs, name, bases, ns)

    def __init__(cls, name, bases, ns, **kwargs):
        super().__init__(name, bases, ns)

    @property
    def __dict__(cls):
        out = {}
        for klass in reversed(cls.__mro__):
            if klass is object:
                continue
            real = type.__dict__["__dict__"].__get__(klass)
            for k, v in _public(real).items():
                out[k] = vars(v) if isinstance(v, Meta) else v
        return out

    def __iter__(cls):
        return iter(vars(cls).items())

    def _update(cls, *dicts, **kw):
        for d in dicts + (kw,):
            for k, v in d.items():
                cur = getattr(cls, k, None)
                if isinstance(cur, Meta) and isinstance(v, dict):
                    cur._update(v)
                else:
                    setattr(cls, k, v)


class ParamsProto(metaclass=Meta):
    pass


class PrefixProto(ParamsProto):
    pass


class Proto:
    def __init__(self, default=None, **kw):
        self.default = default


class Flag(Proto):
    pass
