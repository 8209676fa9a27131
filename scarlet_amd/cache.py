"""Process-global memo ``Cache.check(name, key)`` / ``Cache.set(name, key, value)`` with the
reference's semantics (scarlet/cache.py): ``check`` raises KeyError on a miss."""


class Cache(object):
    _cache = {}

    @staticmethod
    def check(name, key):
        return Cache._cache.setdefault(name, {})[key]

    @staticmethod
    def set(name, key, content):
        Cache._cache.setdefault(name, {})[key] = content

    def __repr__(self):
        return repr(Cache._cache)
