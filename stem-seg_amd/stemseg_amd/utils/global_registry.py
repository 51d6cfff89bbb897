"""Named registries: the reference's plug-in mechanism (stemseg/utils/global_registry.py:23-74).
Heads / backbones are selected by *string* from the config, e.g.
``GlobalRegistry.get("EmbeddingHead")["squeeze_expand_decoder"]`` (model_builder.py:282)."""


class GlobalRegistry(object):
    _all = {}

    def __init__(self, name):
        self._name, self._items = name, {}

    # -- registry of registries -------------------------------------------------------------
    @staticmethod
    def get(name):
        return GlobalRegistry._all.setdefault(name, GlobalRegistry(name))

    @staticmethod
    def exists(name):
        return name in GlobalRegistry._all

    @staticmethod
    def register(registry_name, obj_name=None, obj=None):
        return GlobalRegistry.get(registry_name).add(obj_name, obj)

    # -- one registry -------------------------------------------------------------------------
    def __getitem__(self, key):
        try:
            return self._items[key]
        except KeyError:
            raise KeyError("No object with name '%s' is registered under '%s'" % (key, self._name))

    def __contains__(self, key):
        return key in self._items

    def add(self, name=None, obj=None):
        """``reg.add("x", thing)`` or, as a decorator, ``@reg.add("x")``.  Duplicates are an error
        (global_registry.py:55-58)."""
        def put(n, o):
            n = n or o.__name__
            assert n not in self._items, "An object named '%s' was already registered in '%s' registry!" % (n, self._name)
            self._items[n] = o
            return o
        if obj is not None:
            put(name, obj)
            return None
        return lambda o: put(name, o)
