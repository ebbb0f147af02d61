"""dreammat_amd -- MI355X-native implementation of DreamMat's SDS material-fitting hot path.

Keeps threestudio's plugin mechanism (threestudio/__init__.py:1-13): classes register under the
reference's names and `find(name)` returns them, so `dreammat.yaml` selects them unchanged.
"""
__modules__ = {}


def register(name):
    def decorator(cls):
        __modules__[name] = cls
        return cls

    return decorator


def find(name):
    if name not in __modules__:
        _import_plugins()
    return __modules__[name]


def _import_plugins():
    # importing the modules runs their @register decorators (threestudio/__init__.py:37)
    from . import data, geometry, guidance, material, prompt, renderer, system  # noqa: F401
    from . import background, exporter  # noqa: F401
