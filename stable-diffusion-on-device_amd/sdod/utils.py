"""Module-level lazy properties (public names as in the reference's sdod/utils.py:5-70:
`staticproperty`, `add_module_properties`)."""
import sys
import types


class staticproperty(property):
    """A property whose accessors take no instance argument."""

    @staticmethod
    def _plain(fn):
        return fn.__func__ if isinstance(fn, staticmethod) else fn

    def __init__(self, fget=None, fset=None, fdel=None, doc=None):
        super().__init__(fget, fset, fdel, doc)
        self._sget = None if fget is None else self._plain(fget)
        self._sset = None if fset is None else self._plain(fset)
        self._sdel = None if fdel is None else self._plain(fdel)

    def __get__(self, inst, cls=None):
        if inst is None:
            return self
        if self._sget is None:
            raise AttributeError('unreadable attribute')
        return self._sget()

    def __set__(self, inst, value):
        if self._sset is None:
            raise AttributeError("can't set attribute")
        self._sset(value)

    def __delete__(self, inst):
        if self._sdel is None:
            raise AttributeError("can't delete attribute")
        self._sdel()


class _PropertyModule(types.ModuleType):
    """A module subclass on whose *type* properties can be installed."""
    _props = frozenset()

    def __dir__(self):
        return sorted(set(super().__dir__()) | set(type(self)._props))


def add_module_properties(module_name, properties):
    """Install `properties` ({name: property or callable}) on the module called `module_name`,
    so that `module.name` evaluates the getter lazily at every access."""
    module = sys.modules[module_name]
    cls = type(module)
    if not isinstance(module, _PropertyModule):
        cls = type('_PropertyModule__' + module_name.replace('.', '_'), (_PropertyModule,), {'_props': frozenset()})
    for name, prop in properties.items():
        if not isinstance(prop, property):
            prop = property(prop)
        setattr(cls, name, prop)
        cls._props = cls._props | {name}
    if not isinstance(module, _PropertyModule):
        replacement = cls(module_name)
        replacement.__dict__.update(module.__dict__)
        sys.modules[module_name] = replacement
