"""Small helpers under the names the reference uses (`scoreperformer/utils/functions.py:16-44,91-108`): None tests, defaults,
mask reduction and the string enum whose lookup failure lists the valid values."""
import enum
import functools
import inspect
import operator


def exists(val) -> bool:
    """True for anything but None (0, empty containers and False do exist)."""
    return not (val is None)


def default(val, d):
    """`val` unless it is None; a plain function given as the default is called to produce it (lazily built defaults)."""
    if val is not None:
        return val
    return d() if inspect.isfunction(d) else d


class equals:
    """Predicate object: `equals(3)(x)` is `x == 3`; further call arguments are accepted and ignored (filter callbacks)."""
    __slots__ = ("val",)

    def __init__(self, val):
        self.val = val

    def __call__(self, x, *_, **__):
        return self.val == x


def or_reduce(masks):
    """Element-wise OR of one or more boolean masks."""
    return functools.reduce(operator.or_, masks)


class ExplicitEnum(str, enum.Enum):
    """String-valued enum: members compare equal to their strings; an unknown value names the admissible ones."""

    @classmethod
    def list(cls):
        return [member.value for member in cls]

    @classmethod
    def has_value(cls, value):
        return value in cls._value2member_map_

    @classmethod
    def _missing_(cls, value):
        valid = list(cls._value2member_map_.keys())
        raise ValueError(f"{value} is not a valid {cls.__name__}, please select one of {valid}")
